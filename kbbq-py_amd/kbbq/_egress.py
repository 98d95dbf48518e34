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


def _slabs(bands, outs, step):
    """(serial number, band, its device plane of new qualities, first row, rows) for slabs of `step` reads."""
    k = 0
    for band, out in zip(bands, outs):
        for first in range(0, band['n'], step):
            yield k, band, out, first, min(step, band['n'] - first)
            k += 1


def emit_records(text, base, bands, outs, slab=1 << 18, sink=None):
    """Print the recalibrated records of this rank -- reads base + band['first'] + row of the fastx.NativeFastq `text`,
    new quality characters in the device planes `outs` (one per band) -- to sys.stdout, rendered by the C++ writer in
    slabs.  A binary stdout gets the bytes through the three-stage pipeline above (copy off the device into
    page-locked buffers | rendering into re-used buffers | write(2)); a text-only stdout (StringIO) gets print(),
    like the reference."""
    sys.stdout.flush()
    raw = sink if sink is not None else getattr(sys.stdout, 'buffer', None)      # sink: a binary file of the caller's
    if raw is None:
        for _, band, out, first, m in _slabs(bands, outs, 1 << 20):
            newq = out[first:first + m].cpu().numpy()
            print(text.format_array(base + band['first'] + first, m, newq).tobytes().decode('latin-1'), end='')
        sys.stdout.flush()
        return
    import torch
    from . import _device as dev
    widest = max([band['pitch'] for band in bands] + [16])
    made = [0]

    def page_locked(nbytes):
        made[0] += 1
        return dev.pinned('egress', made[0], max(nbytes, slab * widest))
    staging = Slots(4, page_locked)
    rendered = Slots(4, lambda nbytes: np.empty(nbytes + (nbytes >> 3), dtype=np.uint8))

    def fetch(item):
        k, band, out, first, m = item
        with stage('D2H'), torch.cuda.device(out.device):           # a new thread starts on device 0
            host = staging.get(k, m * band['pitch'])[:m * band['pitch']].view(m, band['pitch'])
            host.copy_(out[first:first + m], non_blocking=True)
            torch.cuda.current_stream().synchronize()
        return k, base + band['first'] + first, m, host.numpy()

    def render(item):
        k, first, m, newq = item
        with stage('format'):
            return text.format_array(first, m, newq, out=lambda nbytes: rendered.get(k, nbytes))

    def write(buf):
        with stage('write'):
            raw.write(memoryview(buf))

    try:
        pipeline(_slabs(bands, outs, slab), fetch, render, write)
    finally:
        dev.release_pinned('egress')
    raw.flush()
    sys.stdout.flush()
