"""Pass-2 egress: new qualities leave the device, become FASTQ text and are written, slab by slab, the three steps
of successive slabs overlapping (device -> pinned host copy | C++ FASTQ writer | write(2)); replaces the per-read
print of recalibrate.py:153-156 for a binary stdout."""
import queue
import threading

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
