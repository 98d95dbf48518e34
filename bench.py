#!/usr/bin/env python3
"""
bench.py -- bases/s recalibrated (covariate accumulate + delta-Q solve + apply) on
synthetic 2x150 bp reads, device-resident, at N GPUs of one node.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over the rank's whole batch:
  K1 accumulate -> (N > 1: RCCL sum-allreduce of the count tables) -> delta-Q solve -> K2 apply.
Workload at N = 1: BASELINE.json configs[1] (50 M reads, 1 read group, Q0-41); weak scaling:
every rank holds --reads reads (configs[3] at N = 8).  Inputs are generated on the device and
are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd'))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
READ_LEN = 150


def cpu_baseline(sample_reads):
    """The CPU oracle (oracle/, a scalar C port of the reference's algorithm) on a bounded
    sample of the same workload, 1 core.  Reported beside the GPU number; not the target."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import oracle as O
    seq, cseq, qual, meta = O.synth(0, sample_reads, sample_reads, 1)
    t0 = time.perf_counter()
    vectors = O.accumulate(seq, cseq, qual, meta, 1, READ_LEN)
    t1 = time.perf_counter()
    dqs = O.get_delta_qs(*vectors)
    t2 = time.perf_counter()
    O.apply(seq, qual, meta, vectors[0], *dqs)
    t3 = time.perf_counter()
    bases = sample_reads * READ_LEN
    return {'value': bases / (t3 - t0), 'unit': 'bases/s', 'cores': 1, 'kind': 'port',
            'sample': '%d synthetic 2x150 reads (%d bases), oracle C port: accumulate %.2fs + '
                      'solve %.2fs + apply %.2fs' % (sample_reads, bases, t1 - t0, t2 - t1, t3 - t2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--reads', type=int, default=50_000_000, help='reads per GPU')
    ap.add_argument('--rgs', type=int, default=1)
    ap.add_argument('--cpu-sample', type=int, default=6_000_000, help='reads in the CPU baseline sample (0 = skip)')
    ap.add_argument('--layout', choices=('pairs', 'reads'), default='pairs',
                    help='device layout of the resident batch: mate-pair rows (304 B per 2 x 150 bp) or one read per row (2 x 160 B)')
    args = ap.parse_args()

    # stdout carries exactly ONE JSON line: library banners (RCCL prints its version to stdout under
    # NCCL_DEBUG=VERSION) are sent to stderr until the result is printed
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus
    # KBBQ_BENCH_REHEARSE=gloo: every rank on the devices that exist (local % count), gloo instead of RCCL --
    # a functional rehearsal of the N > 1 path on a one-GPU box; its numbers mean nothing
    rehearse = os.environ.get('KBBQ_BENCH_REHEARSE', '') == 'gloo'
    if rehearse:
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    use_dist = 'RANK' in os.environ                 # launched by torch.distributed.run (any N, also 1)
    if use_dist:
        if rehearse:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))

    from kbbq import _device as dev
    from kbbq import parallel, recalibrate
    from kbbq.gatk import applybqsr

    n = args.reads
    R, S = args.rgs, READ_LEN
    batch = dev.ReadBatch.synthetic(rank * n, n, world * n, seed=1, nrg=R)
    layout = 'one read per row, pitch %d' % batch.pitch
    if args.layout == 'pairs' and dev.PairBatch.worthwhile(S, batch.pitch):
        # the product's device format for paired reads of one length: set up before the timed region, like the
        # generation itself (inputs are resident in HBM when timing starts)
        batch = dev.PairBatch.from_reads(batch)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        layout = 'mate-pair rows, pitch %d per pair' % batch.pitch
    if R > 1:
        # many read groups: rows ordered by read group (set up once, like the layout above), so that every K1 / K2
        # slice walks only its own rows
        batch = dev.group_by_rg(batch, R)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        layout += ', rows grouped by read group'
    out = torch.empty_like(batch.qual)
    tables = dev.Tables(R, 2 * S)
    ctx = dev.context()

    def step():
        tables.buf.zero_()
        dev.accumulate(batch, tables, check=False)
        parallel.allreduce_tables(tables.buf)
        lut, shape, _, _ = dev.solve(tables)
        dev.apply(batch, lut, shape, out=out, check=False)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.status()                       # raises if a kernel flagged bad input
    ctx.kernel_ms(0, reset=True); ctx.kernel_ms(1, reset=True)
    ctx.timing(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    ctx.timing(False)
    ctx.status()
    k1_ms, k1_n = ctx.kernel_ms(0)
    k2_ms, k2_n = ctx.kernel_ms(1)

    t = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if rehearse else 'cuda')
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        bases_per_rank = n * S
        total_bases = bases_per_rank * world * args.steps
        k1_avg = k1_ms / max(k1_n, 1) * 1e-3
        k2_avg = k2_ms / max(k2_n, 1) * 1e-3
        # algorithmic bytes per launch: K1 reads seq+cseq+qual = 3 B/base;
        # K2 reads seq+qual and writes qual = 3 B/base
        k1_gbs = 3.0 * bases_per_rank / k1_avg / 1e9
        k2_gbs = 3.0 * bases_per_rank / k2_avg / 1e9
        dom, dom_gbs = ('k1_accumulate', k1_gbs) if k1_avg >= k2_avg else ('k2_apply', k2_gbs)
        # HBM bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE x 2 + WRITE_SIZE,
        # collected separately with rocprofv3 --pmc; see profiles/pmc_traffic.json) scaled to this launch
        traffic = None
        try:
            with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as fh:
                pmc = json.load(fh)
                traffic = (pmc['pairs'] if layout.startswith('mate-pair') else pmc)[dom]['hbm_bytes_per_base'] * bases_per_rank
        except Exception:
            pass
        res = {
            'metric': 'bases/sec recalibrated (2x150 bp)', 'value': total_bases / elapsed,
            'unit': 'bases/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'u8', 'data': 'synthetic' + (' (gloo rehearsal, not a measurement)' if rehearse else ''),
            'config': {'workload': '%d synthetic 2x150 bp reads per GPU, %d read group(s), Q0-41, '
                                   'accumulate + solve + apply end-to-end, device-resident' % (n, R),
                       'reads_per_gpu': n, 'read_len': S, 'read_groups': R, 'layout': layout,
                       'parallelism': 'reads sharded x%d, 1 allreduce of count tables' % world},
            'roofline': {'bound': 'hbm', 'kernel': dom, 'achieved': dom_gbs, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': dom_gbs / HBM_PEAK_GBS, 'traffic': traffic},
            'kernels': {'k1_accumulate': {'avg_ms': k1_avg * 1e3, 'launches': k1_n, 'GB/s': k1_gbs,
                                          'frac': k1_gbs / HBM_PEAK_GBS, 'bytes_per_base': 3},
                        'k2_apply': {'avg_ms': k2_avg * 1e3, 'launches': k2_n, 'GB/s': k2_gbs,
                                     'frac': k2_gbs / HBM_PEAK_GBS, 'bytes_per_base': 3},
                        'host_solve_and_sync_ms': elapsed / args.steps * 1e3 - (k1_avg + k2_avg) * 1e3},
        }
        if world == 1 and args.cpu_sample > 0:
            res['cpu_baseline'] = cpu_baseline(args.cpu_sample)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(res), flush=True)
        os.dup2(2, 1)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
