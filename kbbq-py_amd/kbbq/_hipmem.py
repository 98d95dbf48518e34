"""
Device memory, page-locked host buffers, copies and events for the file path WITHOUT PyTorch.

PyTorch is plumbing in this package (device memory, streams, the RCCL process group) and the single-GPU command line
`kbbq recalibrate -f A B` needs none of it that libkbbq_hip's own C ABI does not offer (kbbq_dev_alloc / _free / _zero,
kbbq_dev_copy_async, kbbq_host_alloc, kbbq_event_*: include/kbbq_hip.h) -- but it paid ~1 s of a 2.4 s command for
`import torch`, its HIP context and the code objects of its kernels.  This module is the small part of torch's surface
that kbbq/_device.py, _egress.py and fastx.py touch, over that C ABI: contiguous tensors (slices along the first axis,
reinterpreting views), `empty` / `zeros` / `from_numpy` / `empty_like`, `copy_`, `cpu()`, `numpy()`, `clone()`, and the
`cuda` namespace's device / stream / event handles.  `_device.use_native_memory()` selects it (kbbq.main does, outside a
launcher); everything else -- the tests, bench.py, multi-GPU runs under torch.distributed -- keeps torch.  All work is
enqueued on the kbbq context's own stream.
"""
import contextlib
import ctypes

import numpy as np

from . import _native as N

uint8, int32, int64, float64 = np.dtype(np.uint8), np.dtype(np.int32), np.dtype(np.int64), np.dtype(np.float64)

import threading


class _Current(threading.local):
    """The device the calling THREAD works on (torch's current device is thread-local too: the egress pipeline's worker
    selects its output's device without touching what the main thread reads -- ADVICE r3).  This module is single-device in
    practice (the command line without torch drives one GPU); the kbbq context's calls made from the egress thread are
    copies, an event and a stream synchronisation on the context's one stream, made while the main thread waits in the
    pipeline's last stage and enqueues nothing itself."""
    index = 0

    def __getitem__(self, i):
        return self.index

    def __setitem__(self, i, value):
        self.index = int(value)


_current = _Current()    # _current[0]: kept as the spelling the rest of the module uses


class Device:
    def __init__(self, kind, index=None):
        self.type, self.index = kind, index

    def __eq__(self, other):
        return isinstance(other, Device) and (self.type, self.index) == (other.type, other.index)

    def __hash__(self):
        return hash((self.type, self.index))

    def __repr__(self):
        return 'device(%s)' % (self.type if self.index is None else '%s:%d' % (self.type, self.index))


CPU = Device('cpu')


def _device_of(device):
    if device is None or device == 'cpu' or device is CPU:
        return CPU
    if isinstance(device, Device):
        return device
    if isinstance(device, str) and device.startswith('cuda'):
        return Device('cuda', int(device.split(':')[1]) if ':' in device else _current[0])
    if isinstance(device, int):
        return Device('cuda', device)
    raise ValueError('unknown device %r' % (device,))


def _ctx(index):
    from . import _device
    return _device.context(index)


class _Pool:
    """Accounting of the device memory this module holds, and -- once a reserve limit is set (the streaming file path does:
    kbbq/_stream.py) -- a free list: a released allocation is kept and handed to the next request of about its size instead
    of going back to the runtime (hipFree waits for the device, hipMalloc of a 100 MB slab takes a fraction of a millisecond:
    a path that allocates and frees a slab per step pays both per step).  All work of this module is enqueued on ONE
    stream, so a block released by one step and reused by the next is ordered behind the kernels that still read it."""

    def __init__(self):
        self.lock = threading.Lock()
        self.live = self.peak = self.reserved = self.peak_reserved = 0
        self.limit = None                    # bytes this module may hold (live + kept); None: nothing is kept
        self.keep_all = False                # a process about to end (the command line): nothing goes back to the runtime
        self.kept = []                       # (nbytes, ptr, index)

    def take(self, index, nbytes):
        nbytes = int(nbytes)
        with self.lock:
            best = None
            for k, (size, ptr, idx) in enumerate(self.kept):
                if idx == index and nbytes <= size <= nbytes + (nbytes >> 2) + 4096 and (best is None or size < self.kept[best][0]):
                    best = k
            if best is not None:
                size, ptr, _ = self.kept.pop(best)
                self._count(size, 0)
                return ptr, size
            if self.limit is not None and self.kept and self.reserved + nbytes > self.limit:
                self._release_kept()
        p = ctypes.c_void_p()
        rc = N.load().kbbq_dev_alloc(_ctx(index).handle, nbytes, ctypes.byref(p))
        if rc:                               # out of memory while blocks are being kept: give them back and ask once more
            with self.lock:
                had = bool(self.kept)
                self._release_kept()
            if had:
                rc = N.load().kbbq_dev_alloc(_ctx(index).handle, nbytes, ctypes.byref(p))
        N.check(rc)
        with self.lock:
            self._count(nbytes, nbytes)
        return p.value, nbytes

    def give(self, index, ptr, size):
        with self.lock:
            self.live -= size
            if (self.keep_all and self.limit is None) or (self.limit is not None and size >= (1 << 20) and self.reserved <= self.limit):
                self.kept.append((size, ptr, index))
                return
            self.reserved -= size
        N.load().kbbq_dev_free(_ctx(index).handle, ctypes.c_void_p(ptr))

    def _count(self, live, reserved):
        self.live += live
        self.reserved += reserved
        self.peak = max(self.peak, self.live)
        self.peak_reserved = max(self.peak_reserved, self.reserved)

    def _release_kept(self):                 # lock held
        for size, ptr, idx in self.kept:
            N.load().kbbq_dev_free(_ctx(idx).handle, ctypes.c_void_p(ptr))
            self.reserved -= size
        self.kept = []

    def empty_cache(self):
        with self.lock:
            self._release_kept()


_pool = _Pool()


class _DeviceMemory:
    """One device allocation (kbbq_dev_alloc, or a kept one of the pool), released with the last tensor that views it."""

    def __init__(self, index, nbytes):
        self.index = index
        self.ptr, self.size = _pool.take(index, max(int(nbytes), 16))

    def __del__(self):
        try:
            if self.ptr:
                ptr, self.ptr = self.ptr, None
                _pool.give(self.index, ptr, self.size)
        except Exception:                    # interpreter shutdown: the process's memory goes with it
            pass


class _PinnedMemory:
    """One page-locked host allocation (kbbq_host_alloc)."""

    def __init__(self, nbytes):
        p = ctypes.c_void_p()
        N.check(N.load().kbbq_host_alloc(int(nbytes), ctypes.byref(p)))
        self.ptr = p.value

    def __del__(self):
        try:
            if self.ptr:
                N.load().kbbq_host_free(ctypes.c_void_p(self.ptr))
                self.ptr = None
        except Exception:
            pass


def _count(shape):
    n = 1
    for s in shape:
        n *= int(s)
    return n


class Tensor:
    """A C-contiguous array in device memory, page-locked host memory or a NumPy array's memory."""

    def __init__(self, ptr, shape, dtype, device, owner):
        self._ptr, self.shape, self.dtype, self.device, self._owner = int(ptr), tuple(int(s) for s in shape), np.dtype(dtype), device, owner

    # ---- description
    def data_ptr(self):
        return self._ptr

    def numel(self):
        return _count(self.shape)

    def element_size(self):
        return self.dtype.itemsize

    @property
    def nbytes(self):
        return self.numel() * self.dtype.itemsize

    @property
    def is_cuda(self):
        return self.device.type == 'cuda'

    def __len__(self):
        return self.shape[0]

    # ---- views
    def __getitem__(self, key):
        if isinstance(key, tuple):
            if len(key) == 2 and key[1] == slice(None):
                key = key[0]
            else:
                raise IndexError('_hipmem tensors slice along the first axis only')
        if not isinstance(key, slice) or key.step not in (None, 1):
            raise IndexError('_hipmem tensors take contiguous slices of the first axis')
        lo, hi, _ = key.indices(self.shape[0])
        hi = max(hi, lo)
        row = _count(self.shape[1:]) * self.dtype.itemsize
        return Tensor(self._ptr + lo * row, (hi - lo,) + self.shape[1:], self.dtype, self.device, self._owner)

    def view(self, *shape):
        if len(shape) == 1 and not isinstance(shape[0], (int, np.integer)):
            if isinstance(shape[0], (tuple, list)):
                shape = tuple(shape[0])
            else:                                                    # reinterpret the bytes as another dtype (last axis rescaled)
                dt = np.dtype(shape[0])
                last = self.shape[-1] * self.dtype.itemsize
                if last % dt.itemsize:
                    raise ValueError('view: the last axis does not hold whole elements of %s' % dt)
                return Tensor(self._ptr, self.shape[:-1] + (last // dt.itemsize,), dt, self.device, self._owner)
        shape = list(int(s) for s in shape)
        if -1 in shape:
            k = shape.index(-1)
            shape[k] = self.numel() // max(_count(s for s in shape if s != -1), 1)
        if _count(shape) != self.numel():
            raise ValueError('view: %s does not hold %d elements' % (shape, self.numel()))
        return Tensor(self._ptr, shape, self.dtype, self.device, self._owner)

    reshape = view

    # ---- host access
    def numpy(self):
        if self.is_cuda:
            raise TypeError('numpy() of a device tensor: call cpu() first')
        if isinstance(self._owner, np.ndarray) and self._owner.ctypes.data == self._ptr and self._owner.nbytes == self.nbytes:
            return self._owner.view(self.dtype).reshape(self.shape)
        raw = (ctypes.c_uint8 * max(self.nbytes, 1)).from_address(self._ptr)
        raw._keeps = self._owner                                     # the memory lives as long as the array does
        return np.frombuffer(raw, dtype=np.uint8, count=self.nbytes).view(self.dtype).reshape(self.shape)

    def cpu(self):
        if not self.is_cuda:
            return self
        host = np.empty(self.shape, dtype=self.dtype)
        N.check(N.load().kbbq_dev_download(_ctx(self.device.index).handle, N.ptr(host), ctypes.c_void_p(self._ptr), self.nbytes))
        return from_numpy(host)

    def to(self, device):
        dev = _device_of(device)
        if dev == self.device:
            return self
        if dev.type == 'cpu':
            return self.cpu()
        out = empty(self.shape, dtype=self.dtype, device=dev)
        out.copy_(self)
        return out

    def cuda(self):
        return self.to('cuda')

    def clone(self):
        out = empty(self.shape, dtype=self.dtype, device=self.device)
        out.copy_(self)
        return out

    # ---- contents
    def copy_(self, src, non_blocking=False):
        if isinstance(src, np.ndarray):
            src = from_numpy(np.ascontiguousarray(src))
        if src.nbytes != self.nbytes:
            raise ValueError('copy_: %d bytes into %d' % (src.nbytes, self.nbytes))
        if not self.is_cuda and not src.is_cuda:
            ctypes.memmove(self._ptr, src._ptr, self.nbytes)
            return self
        index = self.device.index if self.is_cuda else src.device.index
        ctx = _ctx(index)
        kind = 3 if (self.is_cuda and src.is_cuda) else (1 if self.is_cuda else 2)
        N.check(N.load().kbbq_dev_copy_async(ctx.handle, ctypes.c_void_p(self._ptr), ctypes.c_void_p(src._ptr), self.nbytes, kind))
        if not non_blocking or not (isinstance(self._owner, _PinnedMemory) or isinstance(src._owner, _PinnedMemory) or kind == 3):
            ctx.sync()                       # pageable host memory: the copy must have left it before the caller goes on
        return self

    def zero_(self):
        if self.is_cuda:
            N.check(N.load().kbbq_dev_zero(_ctx(self.device.index).handle, ctypes.c_void_p(self._ptr), self.nbytes))
        else:
            ctypes.memset(self._ptr, 0, self.nbytes)
        return self


def empty(*shape, dtype=np.float32, device=None, pin_memory=False):
    if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
        shape = tuple(shape[0])
    dt, dev = np.dtype(dtype), _device_of(device)
    nbytes = _count(shape) * dt.itemsize
    if dev.type == 'cuda':
        mem = _DeviceMemory(dev.index, nbytes)
        return Tensor(mem.ptr, shape, dt, dev, mem)
    if pin_memory:
        mem = _PinnedMemory(nbytes)
        return Tensor(mem.ptr, shape, dt, CPU, mem)
    arr = np.empty(shape, dtype=dt)
    return Tensor(arr.ctypes.data, shape, dt, CPU, arr)


def zeros(*shape, dtype=np.float32, device=None):
    return empty(*shape, dtype=dtype, device=device).zero_()


def empty_like(t):
    return empty(t.shape, dtype=t.dtype, device=t.device)


def from_numpy(arr):
    if not arr.flags['C_CONTIGUOUS']:
        raise ValueError('from_numpy: a C-contiguous array is needed')
    return Tensor(arr.ctypes.data, arr.shape, arr.dtype, CPU, arr)


def equal(a, b):
    return a.shape == b.shape and np.array_equal(a.cpu().numpy(), b.cpu().numpy())


class _Stream:
    cuda_stream = None                       # the kbbq context's own stream: _device.context() leaves it in place

    def __init__(self, index):
        self.index = index

    def synchronize(self):
        _ctx(self.index).sync()


class Event:
    def __init__(self, enable_timing=False):
        self._h, self._index = None, _current[0]

    def record(self):
        ctx = _ctx(self._index)
        if self._h is None:
            h = ctypes.c_void_p()
            N.check(N.load().kbbq_event_create(ctx.handle, ctypes.byref(h)))
            self._h = h
        N.check(N.load().kbbq_event_record(ctx.handle, self._h))

    def synchronize(self):
        if self._h is not None:
            N.check(N.load().kbbq_event_sync(self._h))

    def __del__(self):
        try:
            if self._h is not None:
                N.load().kbbq_event_destroy(self._h)
        except Exception:
            pass


class _Cuda:
    Event = Event

    @staticmethod
    def is_available():
        n = ctypes.c_int(0)
        return N.load().kbbq_device_count(ctypes.byref(n)) == N.KBBQ_OK and n.value > 0

    @staticmethod
    def device_count():
        n = ctypes.c_int(0)
        N.load().kbbq_device_count(ctypes.byref(n))
        return n.value

    @staticmethod
    def mem_get_info(index=None):
        """(free, total) bytes of the device, as torch.cuda.mem_get_info."""
        f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)
        N.check(N.load().kbbq_dev_mem_info(_ctx(_current[0] if index is None else getattr(index, 'index', index)).handle,
                                           ctypes.byref(f), ctypes.byref(t)))
        return f.value, t.value

    @staticmethod
    def max_memory_allocated(index=None):
        return _pool.peak

    @staticmethod
    def max_memory_reserved(index=None):
        return _pool.peak_reserved

    @staticmethod
    def memory_allocated(index=None):
        return _pool.live

    @staticmethod
    def reset_peak_memory_stats(index=None):
        _pool.peak, _pool.peak_reserved = _pool.live, _pool.reserved

    @staticmethod
    def empty_cache():
        _pool.empty_cache()

    @staticmethod
    def set_reserve_limit(nbytes):
        """From now on released device allocations are kept for reuse while the module holds no more than `nbytes` (None:
        back to releasing everything at once)."""
        _pool.limit = None if nbytes is None else int(nbytes)
        if nbytes is None and not _pool.keep_all:
            _pool.empty_cache()

    @staticmethod
    def keep_released_memory(on=True):
        """Released device allocations stay with this module (to be reused, or to go with the process): hipFree waits for the
        device and unmaps -- 10-20 ms per GB-sized plane, 80 ms at the end of the command line for memory the exiting
        process returns anyway.  An allocation that fails gives everything kept back first (_Pool.take)."""
        _pool.keep_all = bool(on)

    @staticmethod
    def current_device():
        return _current[0]

    @staticmethod
    def set_device(index):
        _current[0] = int(getattr(index, 'index', index) or 0)

    @staticmethod
    @contextlib.contextmanager
    def device(dev):
        index = dev if isinstance(dev, int) else getattr(dev, 'index', None)
        saved = _current[0]
        if index is not None:
            _current[0] = index
        try:
            yield
        finally:
            _current[0] = saved

    @staticmethod
    def current_stream(index=None):
        return _Stream(_current[0] if index is None else getattr(index, 'index', index))

    @staticmethod
    def synchronize(index=None):
        _ctx(_current[0] if index is None else getattr(index, 'index', index)).sync()



cuda = _Cuda()
