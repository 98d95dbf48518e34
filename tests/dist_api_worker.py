"""Launched by tests under torch.distributed.run: kbbq.gatk.bqsr.bam_to_bqsr_covariates on every rank; rank 0 writes
the nine vectors (or the exception every rank raised) as JSON to the path given last."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd'))

from kbbq import aln, benchmark, parallel   # noqa: E402
from kbbq.gatk import bqsr                  # noqa: E402

class ReadObjects:
    """The alignments as read objects behind a pysam-like file object (header + iteration): the non-native path."""

    def __init__(self, f):
        self._f, self.header = f, f.header

    def __iter__(self):
        return iter(list(self._f))


if __name__ == '__main__':
    sam, fa, vcf, native, out = sys.argv[1:6]
    world, rank = parallel.init_from_env()
    reads = aln.AlignmentFile(sam)
    try:
        vec = bqsr.bam_to_bqsr_covariates(reads if native == '1' else ReadObjects(reads), fa, benchmark.get_var_sites(vcf))
        res = {'vectors': [v.tolist() for v in vec]}
    except Exception as e:                   # noqa: BLE001 -- reported, then re-raised for the exit code
        res = {'error': type(e).__name__}
        if rank == 0:
            json.dump(res, open(out, 'w'))
        raise
    if rank == 0:
        json.dump(res, open(out, 'w'))
