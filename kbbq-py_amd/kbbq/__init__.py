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

from . import compare_reads
from . import fastx
