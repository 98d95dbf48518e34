#!/usr/bin/env python3
"""
kbbq command line -- the `recalibrate` sub-command of the reference CLI
and the `benchmark` sub-command (reference kbbq/main.py:26-89).  `plot` is out of scope here.
"""
import argparse

from . import __version__
from . import recalibrate as _recal


_ENDS_WITH_THE_COMMAND = False               # set by `python -m kbbq.main`: the process ends (main._leave) when the command has run


def recalibrate(args):
    import os
    from . import parallel
    world, _ = parallel.init_from_env()          # one process per GPU under torch.distributed.run; no-op otherwise
    if world == 1 and 'torch' not in __import__('sys').modules and not os.environ.get('KBBQ_USE_TORCH'):
        # one GPU: nothing of PyTorch is needed -- device memory, page-locked slabs, copies and events come from the library's
        # own C ABI (kbbq/_hipmem.py) and `import torch` (~1 s with its HIP context) never happens
        from . import _device, _hipmem, fastx
        _device.use_native_memory()
        if _ENDS_WITH_THE_COMMAND and not os.environ.get('KBBQ_SLOW_EXIT'):
            # this process ends right after its last byte (_leave): what it holds goes with it instead of being returned piece by piece
            _hipmem.cuda.keep_released_memory(True)
            fastx.LEAVE_OPEN = True
    _recal.recalibrate(bam=args.bam, fastq=args.fastq, infer_rg=args.infer_rg,
                       use_oq=args.use_oq, set_oq=args.set_oq, gatkreport=args.gatkreport, output=args.output)


def benchmark(args):
    from . import benchmark as _bm
    from . import parallel
    parallel.init_from_env()          # one process per GPU under torch.distributed.run; no-op otherwise
    _bm.benchmark(bamfile=args.bam, fafile=args.reference, vcffile=args.vcf, fastqfile=args.fastq,
                  label=args.label, use_oq=args.use_oq, bedfh=args.bedfile)


def main():
    parser = argparse.ArgumentParser(description='K-mer Based Base Quality score recalibration (MI355X build)')
    parser.add_argument('-v', '--version', action='version', version=__version__)
    sub = parser.add_subparsers(title='command', description='valid commands')
    parser.set_defaults(command=lambda a: parser.print_help)
    sub.add_parser('help', description='Print help information').set_defaults(
        command=lambda a: parser.print_help)

    rp = sub.add_parser('recalibrate', description='Recalibrate a BAM or FASTQ file')
    src = rp.add_mutually_exclusive_group(required=True)
    src.add_argument('-b', '--bam', help='BAM to recalibrate')
    src.add_argument('-f', '--fastq', nargs=2,
                     help='FASTQ file to recalibrate and an error-corrected version of it.')
    rp.add_argument('-u', '--use-oq', action='store_true',
                    help='Use the OQ tag for quality scores (BAM input only).')
    rp.add_argument('-s', '--set-oq', action='store_true',
                    help="Set the 'OQ' tag before recalibration (BAM output only).")
    rp.add_argument('-g', '--gatkreport', help='Load the model from / save it to a GATK report.')
    rp.add_argument('-o', '--output', default=None,
                    help='Write the recalibrated FASTQ to this file instead of stdout (not in the reference); under '
                         'torch.distributed.run every rank writes FILE.rankNNNN, to be concatenated in rank order.')
    rp.add_argument('--infer-rg', action='store_true',
                    help='Infer the read group from the FASTQ read name (name_RG:Z:id).')
    rp.set_defaults(command=recalibrate)

    bp = sub.add_parser('benchmark', description='Benchmark a SAM or FASTQ file using a truth set')
    req = bp.add_argument_group(title='required arguments')
    req.add_argument('-b', '--bam', required=True,
                     help='Truth set alignments (SAM text). Differences from the reference at nonvariable sites are errors.')
    req.add_argument('-r', '--reference', required=True, help='FASTA file containing the reference genome')
    req.add_argument('-v', '--vcf', required=True, help='VCF file containing variable sites')
    bp.add_argument('-f', '--fastq', default=None, help='fastq file to benchmark')
    bp.add_argument('-l', '--label', default=None, help='label to use for label column')
    bp.add_argument('-u', '--use-oq', action='store_true', help='Use the OQ tag for quality scores')
    bp.add_argument('-d', '--bedfile', type=argparse.FileType('r'),
                    help='BED file of confident regions. Sites outside the given regions will be skipped.')
    bp.set_defaults(command=benchmark)

    args = parser.parse_args()
    args.command(args)


def _leave():
    """The command's last step when it ran as a single process without torch: flush, run the exit handlers (the stage report of
    KBBQ_TIMING) and end the process without taking it apart piece by piece -- unmapping 5 GB of input, returning a gigabyte of
    page-locked buffers and shutting the HIP runtime down cost the command 0.1-0.2 s after its last byte was written; the
    kernel releases all of it faster.  Under a launcher (torch imported: process group, RCCL) the ordinary exit stays."""
    import atexit
    import os
    import sys
    if 'torch' in sys.modules or os.environ.get('KBBQ_SLOW_EXIT'):
        return
    code = 0
    try:
        sys.stdout.flush()
        sys.stderr.flush()
        atexit._run_exitfuncs()
        sys.stdout.flush()
        sys.stderr.flush()
    except BaseException:                    # noqa: BLE001 -- a flush that fails (closed pipe, full disk) must not look like success:
        code = 120                           # the status the interpreter itself leaves with when its final flush fails
    os._exit(code)


if __name__ == '__main__':
    _ENDS_WITH_THE_COMMAND = True
    main()
    _leave()
