"""
The host side of one process per GPU (csrc/host_threads.h, csrc/host_affinity.cpp; no GPU): a rank of a node starts
1 / LOCAL_WORLD_SIZE of the usable CPUs' worth of host threads -- eight ranks together no more than one process alone --
and binds them to the NUMA node of its GPU.  Every case runs in a fresh process: the usable-CPU figure is cached per
process and binding changes the caller's affinity mask.
"""
import json
import os
import subprocess
import sys

from conftest import ROOT

PROBE = r"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.join(%r, 'kbbq-py_amd'))
from kbbq import _native as N
lib = N.load()
out = {'threads': lib.kbbq_host_threads(1 << 40), 'small': lib.kbbq_host_threads(100), 'before': sorted(os.sched_getaffinity(0))}
if len(sys.argv) > 1:
    node, ncpus = ctypes.c_int(-7), ctypes.c_int(-7)
    out['rc'] = lib.kbbq_bind_host_to_pci(sys.argv[1].encode(), ctypes.byref(node), ctypes.byref(ncpus))
    out.update(node=node.value, ncpus=ncpus.value, after=sorted(os.sched_getaffinity(0)), threads_after=lib.kbbq_host_threads(1 << 40))
print(json.dumps(out))
""" % ROOT


def _probe(env=None, *argv):
    e = {k: v for k, v in os.environ.items() if k not in ('LOCAL_WORLD_SIZE', 'KBBQ_LOCAL_RANKS', 'KBBQ_HOST_THREADS', 'KBBQ_SYSFS_ROOT')}
    e.update(env or {})
    r = subprocess.run([sys.executable, '-c', PROBE] + list(argv), env=e, capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return json.loads(r.stdout.decode().strip().splitlines()[-1])


def test_eight_ranks_start_no_more_threads_than_one_process():
    alone = _probe()
    usable = alone['threads']
    assert usable >= 1 and alone['small'] == 1                   # little work: one thread whatever the ceiling
    for ranks in (2, 4, 8):
        per_rank = [_probe({'LOCAL_WORLD_SIZE': str(ranks)})['threads'] for _ in range(2)]      # every rank computes the same share
        assert per_rank[0] == per_rank[1] == max(1, usable // ranks)
        assert per_rank[0] * ranks <= max(usable, ranks)        # in total: the usable CPUs (one thread each when there are fewer CPUs than ranks)
    # another launcher's variable wins over the launcher's, an explicit ceiling over both
    assert _probe({'LOCAL_WORLD_SIZE': '8', 'KBBQ_LOCAL_RANKS': '1'})['threads'] == usable
    assert _probe({'LOCAL_WORLD_SIZE': '8', 'KBBQ_HOST_THREADS': '5'})['threads'] == 5
    assert _probe({'LOCAL_WORLD_SIZE': 'zero'})['threads'] == usable


def _sysfs(tmp_path, node, cpulist, busid='0000:c1:00.0'):
    d = tmp_path / 'bus' / 'pci' / 'devices' / busid
    d.mkdir(parents=True)
    (d / 'numa_node').write_text('%d\n' % node)
    if node >= 0:
        n = tmp_path / 'devices' / 'system' / 'node' / ('node%d' % node)
        n.mkdir(parents=True)
        (n / 'cpulist').write_text(cpulist + '\n')
    return {'KBBQ_SYSFS_ROOT': str(tmp_path)}


def test_a_rank_binds_its_threads_to_the_numa_node_of_its_gpu(tmp_path):
    have = sorted(os.sched_getaffinity(0))
    if len(have) < 2:
        import pytest
        pytest.skip('one CPU: nothing to narrow')
    half = have[:len(have) // 2]
    cpulist = ','.join(str(c) for c in half) + ',4000-4003'      # CPUs the process may not use anyway are ignored
    env = dict(_sysfs(tmp_path / 'a', 1, cpulist), LOCAL_WORLD_SIZE='2')
    got = _probe(env, '0000:C1:00.0')                          # HIP prints the address in upper case, sysfs in lower case
    assert got['rc'] == 0 and got['node'] == 1 and got['after'] == half and got['ncpus'] == len(half)
    # the share of the CPUs was fixed before the mask narrowed: binding does not halve it a second time
    assert got['threads_after'] == got['threads'] == max(1, len(have) // 2)
    # no node named (single-socket hosts, virtual machines), an unknown device, a node without usable CPUs: nothing changes
    for env, bus in ((_sysfs(tmp_path / 'b', -1, ''), '0000:c1:00.0'), (_sysfs(tmp_path / 'c', 0, '0-1'), '0000:99:00.0'),
                     (_sysfs(tmp_path / 'd', 2, '4000-4003'), '0000:c1:00.0')):
        got = _probe(env, bus)
        assert got['rc'] == 0 and got['node'] == -1 and got['after'] == got['before'] and got['ncpus'] == len(have)
    # ranges and lists in one cpulist
    env = _sysfs(tmp_path / 'e', 0, '%d-%d' % (have[0], have[-1]))
    got = _probe(env, '0000:c1:00.0')
    assert got['node'] == 0 and got['after'] == have
