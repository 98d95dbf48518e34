"""
kbbq -- MI355X-native drop-in for the recalibrate hot path of adamjorr/kbbq-py.

Same module and function names as the reference for that path
(kbbq.recalibrate, kbbq.compare_reads, kbbq.covariate, kbbq.read,
kbbq.gatk.applybqsr, kbbq.main); the per-base work runs in hand-written HIP
kernels (libkbbq_hip.so, C ABI in include/kbbq_hip.h).  The benchmark / plot /
GATK-report subsystems of the reference are out of scope.
"""
__all__ = ['compare_reads', 'recalibrate', 'covariate', 'read', 'fastx', 'benchmark', 'aln']
__version__ = '0.0.0'


def _start_the_device_early():
    """`python -m kbbq.main recalibrate ...` as a single process: the HIP runtime's initialisation (0.10-0.12 s inside the first call of
    the library) starts NOW, on a thread of its own, while this thread still imports NumPy and the package's modules (0.09 s) -- it used
    to start after them.  Anything else that imports the package (tests, a launcher's ranks, other commands) is left alone."""
    import os
    import sys
    if sys.argv[:2] != ['-m', 'recalibrate'] or 'RANK' in os.environ or os.environ.get('KBBQ_USE_TORCH') or os.environ.get('KBBQ_LATE_DEVICE'):
        return
    import threading

    def run():
        try:
            import ctypes
            from . import _native
            _native.load().kbbq_device_count(ctypes.byref(ctypes.c_int(0)))
        except Exception:                    # noqa: BLE001 -- whatever is wrong is reported where the device is first needed
            pass
    threading.Thread(target=run, daemon=True).start()


_start_the_device_early()

from . import compare_reads
from . import fastx
