"""Stage timing of the file path (KBBQ_TIMING=1: one line per stage on stderr when the command ends)."""
import atexit
import contextlib
import os
import sys
import threading
import time

ON = bool(os.environ.get('KBBQ_TIMING'))
TIMELINE = os.environ.get('KBBQ_TIMING') == '2'      # also every stage's start / end since the process started
_acc = {}
_events = []
_lock = threading.Lock()


def _process_start():
    try:
        import psutil
        return psutil.Process().create_time()
    except Exception:
        return _imported


_imported = time.time()


@contextlib.contextmanager
def stage(name, sync=False):
    if not ON:
        yield
        return
    t0, w0 = time.perf_counter(), time.time()
    try:
        yield
    finally:
        if sync:
            from . import _device
            _device.synchronize()
        with _lock:                       # several threads may close the same stage (the positioned writers)
            s, k = _acc.get(name, (0.0, 0))
            _acc[name] = (s + time.perf_counter() - t0, k + 1)
        if TIMELINE:
            _events.append((w0, time.time(), name))


def collect(enable):
    """Programmatic form of KBBQ_TIMING (bench.py): collect(True) starts collecting stage times in this process,
    collect(False) stops and returns {stage: seconds} of what was collected."""
    global ON
    if enable:
        _acc.clear()
        ON = True
        return None
    out = {n: round(s, 4) for n, (s, _) in _acc.items()}
    _acc.clear()
    ON = bool(os.environ.get('KBBQ_TIMING'))
    return out


def report(reset=True):
    if ON and _acc:
        total = sum(s for s, _ in _acc.values())
        sys.stderr.write('kbbq stages: ' + '  '.join('%s %.3fs' % (n, s) for n, (s, _) in _acc.items())
                         + '  (sum %.3fs)\n' % total)
        try:
            from .parallel import HOST_BINDING as hb
            if hb:
                sys.stderr.write('kbbq host: %d host threads (1 / %d of the CPUs the job may use), NUMA node %s, %d CPUs in the mask\n'
                                 % (hb['host_threads'], hb['local_ranks'], hb['numa_node'] if hb['numa_node'] >= 0 else 'not named', hb['cpus']))
        except Exception:
            pass
        if TIMELINE:
            t0 = _process_start()
            sys.stderr.write('kbbq timeline (s since the process started; this module was imported at %.3f, now %.3f):\n'
                             % (_imported - t0, time.time() - t0))
            for a, b, n in sorted(_events):
                if b - a >= 0.002:
                    sys.stderr.write('  %7.3f .. %7.3f  %s\n' % (a - t0, b - t0, n))
        if reset:
            _acc.clear()
            del _events[:]


atexit.register(report)
