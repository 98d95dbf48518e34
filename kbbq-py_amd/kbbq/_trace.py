"""Stage timing of the file path (KBBQ_TIMING=1: one line per stage on stderr when the command ends)."""
import atexit
import contextlib
import os
import sys
import time

ON = bool(os.environ.get('KBBQ_TIMING'))
_acc = {}


@contextlib.contextmanager
def stage(name, sync=False):
    if not ON:
        yield
        return
    t0 = time.perf_counter()
    try:
        yield
    finally:
        if sync:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        s, k = _acc.get(name, (0.0, 0))
        _acc[name] = (s + time.perf_counter() - t0, k + 1)


def report(reset=True):
    if ON and _acc:
        total = sum(s for s, _ in _acc.values())
        sys.stderr.write('kbbq stages: ' + '  '.join('%s %.3fs' % (n, s) for n, (s, _) in _acc.items())
                         + '  (sum %.3fs)\n' % total)
        if reset:
            _acc.clear()


atexit.register(report)
