"""Pass-2 egress: new qualities leave the device, become FASTQ text and are written, slab by slab, the three steps
of successive slabs overlapping (device -> pinned host copy | C++ FASTQ writer | write(2)); replaces the per-read
print of recalibrate.py:153-156 for a binary stdout."""
import queue
import sys
import threading

import numpy as np

from ._trace import stage

_END = object()


def pipeline(source, *stages, depth=1):
    """Run items of `source` through `stages` (callables item -> item), one thread per stage except the last, which
    runs here; bounded queues give back-pressure (at most len(stages) + depth * (len(stages) - 1) items are between
    the start of the first stage and the end of the last).  The first exception of any stage is re-raised here after
    every thread has drained (no stage is left blocked)."""
    qs = [queue.Queue(depth) for _ in stages]
    errors, stop = [], threading.Event()

    def feed():
        try:
            for item in source:
                if stop.is_set():
                    break
                qs[0].put(item)
        except BaseException as e:           # noqa: BLE001 -- re-raised by the caller's thread
            errors.append(e)
            stop.set()
        finally:
            qs[0].put(_END)

    def work(i):
        while True:
            item = qs[i].get()
            if item is _END:
                break
            if stop.is_set():
                continue                     # keep draining so that nobody upstream blocks
            try:
                res = stages[i](item)
            except BaseException as e:       # noqa: BLE001
                errors.append(e)
                stop.set()
                continue
            if i + 1 < len(stages):
                qs[i + 1].put(res)
        if i + 1 < len(stages):
            qs[i + 1].put(_END)

    threads = [threading.Thread(target=feed, daemon=True)]
    threads += [threading.Thread(target=work, args=(i,), daemon=True) for i in range(len(stages) - 1)]
    for t in threads:
        t.start()
    work(len(stages) - 1)
    for t in threads:
        t.join()
    if errors:
        raise errors[0]


class Slots:
    """Round-robin buffers for a pipeline stage: slot k is reused by item k + count, which the pipeline's bound on
    items in flight keeps from starting before item k has left the stages that read the buffer."""

    def __init__(self, count, make):
        self._bufs = [None] * count
        self._make = make

    def get(self, k, nbytes):
        i = k % len(self._bufs)
        b = self._bufs[i]
        if b is None or b.shape[0] < nbytes:
            b = self._bufs[i] = self._make(nbytes)
        return b


class _Reserve:
    """The blocks of every slab are reserved (fallocate, without moving the file's end) just before the slab is written: on disk-backed
    filesystems write(2) into the page cache then runs 15 % faster (0.27 -> 0.23 s per 2.5 GB; neutral on tmpfs: scripts/gpu_write_bench.sh).
    Regular files only; the first refusal (a filesystem without fallocate) ends it.  KBBQ_FALLOCATE=0 turns it off."""

    def __init__(self, raw):
        import os
        import stat
        self.fd, self.at, self.call = None, 0, None
        if os.environ.get('KBBQ_FALLOCATE') == '0':
            return
        try:
            fd = raw.fileno()
            if not stat.S_ISREG(os.fstat(fd).st_mode):
                return
            import ctypes
            libc = ctypes.CDLL(None, use_errno=True)
            libc.fallocate.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_longlong]
            libc.fallocate.restype = ctypes.c_int
            raw.flush()
            self.at = os.lseek(fd, 0, os.SEEK_CUR)
            self.fd, self.call = fd, libc.fallocate
        except Exception:                    # noqa: BLE001 -- an optimisation: anything unusual about the sink and it stays out of the way
            self.fd = None

    def ahead(self, nbytes):
        if self.fd is not None and self.call(self.fd, 1, self.at, nbytes) != 0:      # 1 = FALLOC_FL_KEEP_SIZE
            self.fd = None
        self.at += nbytes


def _slabs(produced, step):
    """(serial number, band, its device plane of new qualities, first read, reads) for slabs of `step` reads (even, so
    that a slab of mate-pair rows starts at a first mate).  produced: (band, plane) pairs -- a list, or a generator that
    makes the planes as it is asked for them (the streaming file path: kbbq/_stream.py)."""
    k = 0
    for band, out in produced:
        for first in range(0, band['n'], step):
            yield k, band, out, first, min(step, band['n'] - first)
            k += 1


def _rows_of(band, first, m):
    """Rows [r0, r1) of a band's output plane that hold reads [first, first + m) (first even): the band's own layout --
    two reads to a mate-pair row (recalibrate.apply_band's out_flags), else one read per row."""
    if band.get('out_flags', 0) & 1:
        return first // 2, (first + m + 1) // 2
    return first, first + m


def emit_records(text, base, bands, outs, slab=1 << 18, sink=None):
    """Print the recalibrated records of this rank -- reads base + band['first'] + i of the fastx.NativeFastq `text`,
    new quality characters in the device planes `outs` (one per band, in the layout K2 wrote: mate-pair rows are read as
    they are, kbbq_fastq_format_rows) -- to sys.stdout, rendered by the C++ writer in slabs.  A binary stdout gets the
    bytes through the three-stage pipeline above (copy off the device into page-locked buffers | rendering into re-used
    buffers | write(2)); a text-only stdout (StringIO) gets print(), like the reference."""
    emit_produced(text, base, zip(bands, outs), max([band['pitch'] for band in bands] + [16]), slab, sink)


def emit_produced(text, base, produced, widest, slab=1 << 18, sink=None):
    """emit_records for (band, plane) pairs that are PRODUCED while the earlier ones are being copied, rendered and written
    (a generator: it runs on the pipeline's feeding thread): the streaming file path's pass 2, fill | H2D | K2 | D2H |
    format | write.  A band may carry its own reader (band['text']: a segment of a sequentially read input) and its own
    `base`; widest: the largest row pitch any band will have."""
    sys.stdout.flush()
    raw = sink if sink is not None else getattr(sys.stdout, 'buffer', None)      # sink: a binary file of the caller's
    if raw is None:
        for _, band, out, first, m in _slabs(produced, 1 << 20):
            r0, r1 = _rows_of(band, first, m)
            newq = out[r0:r1].cpu().numpy()
            print(band.get('text', text).format_rows_array(band.get('base', base) + band['first'] + first, m, newq, band.get('out_flags', 0),
                                                           band.get('out_S2', 0)).tobytes().decode('latin-1'), end='')
        sys.stdout.flush()
        return
    from . import _device as dev
    torch = dev._torch()
    made = [0]

    def page_locked(nbytes):
        made[0] += 1
        return dev.pinned('egress', made[0], max(nbytes, slab * widest))
    staging = Slots(4, page_locked)
    def fresh(nbytes):
        buf = np.empty(nbytes + (nbytes >> 3), dtype=np.uint8)
        dev.N.load().kbbq_host_advise_huge(buf.ctypes.data, buf.nbytes)      # 90 MB the rendering threads touch for the first time
        return buf
    rendered = Slots(4, fresh)

    def fetch(item):
        k, band, out, first, m = item
        r0, r1 = _rows_of(band, first, m)
        width = out.shape[1]
        with stage('D2H'), torch.cuda.device(out.device):           # a new thread starts on device 0
            host = staging.get(k, (r1 - r0) * width)[:(r1 - r0) * width].view(r1 - r0, width)
            host.copy_(out[r0:r1], non_blocking=True)
            done = torch.cuda.Event()                                # this copy only: the producer may have enqueued the next
            done.record()                                            # slab's uploads and kernels behind it meanwhile
            done.synchronize()
        return (k, band.get('text', text), band.get('base', base) + band['first'] + first, m, host.numpy(), band.get('out_flags', 0),
                band.get('out_S2', 0))

    def render(item):
        k, reader, first, m, newq, flags, S2 = item
        with stage('format'):
            return reader.format_rows_array(first, m, newq, flags, S2, out=lambda nbytes: rendered.get(k, nbytes))

    reserve = _Reserve(raw)

    def write(buf):
        with stage('write'):
            reserve.ahead(len(buf))
            raw.write(memoryview(buf))

    try:
        pipeline(_slabs(produced, slab), fetch, render, write)
    finally:
        dev.release_pinned('egress')
    raw.flush()
    sys.stdout.flush()
