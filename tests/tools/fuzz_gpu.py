#!/usr/bin/env python3
"""Randomised differential campaign: HIP path (every layout: rows, mate-pair rows, rows grouped by read group, length
bands with and without the shortest-read promise) against the CPU oracle, for --seconds of random shapes.  Prints one
line per 50 cases and every mismatch; exit code 1 if there was one.  Test infrastructure (uses oracle/)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
ap = argparse.ArgumentParser(); ap.add_argument('--seconds', type=float, default=300); ap.add_argument('--seed', type=int, default=1)
a = ap.parse_args()
import numpy as np, torch
import oracle as O
from kbbq import _device as dev, fastx
rng = np.random.default_rng(a.seed)
t_end = time.time() + a.seconds
cases = bad = 0
def check(tag, cond, info):
    global bad
    if not cond:
        bad += 1
        print('MISMATCH %s %s' % (tag, info), flush=True)
while time.time() < t_end:
    cases += 1
    uniform = rng.random() < 0.6
    S = int(rng.choice([int(rng.integers(1, 330)), 150, 151, 100, 16, 17, 48, 250, 300]))
    lo = S if uniform else int(rng.integers(1, S + 1))
    nrg = int(rng.choice([1, 1, 2, 3, 8, int(rng.integers(1, 40))]))
    minscore = int(rng.choice([6, 6, 6, 0, 2, 3, 10, 20, 40]))
    n = int(rng.choice([2, 64, 130, int(rng.integers(2, 6000)), int(rng.integers(2, 60000))])) // 2 * 2
    single_end = rng.random() < 0.25                   # no read is second in pair (lay_out: two reads to a row), any count
    if single_end and rng.random() < 0.5:
        n += 1
    qlo, qhi = int(rng.integers(0, 10)), int(rng.integers(20, 43))
    info = dict(S=S, lo=lo, nrg=nrg, minscore=minscore, n=n, q=(qlo, qhi), seed=cases)
    b = dev.ReadBatch.synthetic(0, n, n, seed=cases, len_lo=lo, len_hi=S, nrg=nrg, qlo=qlo, qhi=qhi)
    if single_end:
        b.meta.bitwise_and_(0x7FFFFFFF)
        info['single_end'] = True
    meta = b.meta[:n].cpu().numpy().view(np.uint32)
    lens = (meta & 0xFFFF).astype(np.int64)
    host = [x.cpu().numpy() for x in (b.seq[:n], b.cseq[:n], b.qual[:n])]
    want = O.accumulate(host[0], host[1], host[2], meta, nrg, S, minscore=minscore)
    dqs = O.get_delta_qs(*want)
    want_q = O.apply(host[0], host[2], meta, want[0], *dqs, minscore=minscore)
    layouts = [('rows', b)]
    if uniform and S >= 2:
        try:
            layouts.append(('pairs', dev.PairBatch.from_reads(b)))
        except ValueError:
            pass
    for name, src in list(layouts):
        layouts.append((name + '+grouped', dev.group_by_rg(src, nrg)))
    # the one-pass native layout (pairs when they apply + read-group gather + 4-bit sequence planes), and the same
    # with character planes; `restore` stores K2's output through the permutation (no ungroup pass)
    for packed in (True, False):
        laid = dev.lay_out(b, nrg, S, packed=packed)
        if laid is not b:
            layouts.append(('lay_out(packed=%s)' % packed, laid))
    lut = shape = None
    for name, lay in layouts:
        t = dev.Tables(nrg, 2 * S)
        try:
            dev.accumulate(lay, t, minscore)
        except dev.N.LutNeedsCheckedApply:
            continue                                   # a shape this layout does not serve (the product falls back to rows)
        for k, (got, w) in enumerate(zip(t.to_host(), want[5:9])):
            check('tables[%d] %s' % (k, name), np.array_equal(got, w), info)
        lut, shape, _, _ = dev.solve(t, minscore=minscore)
        restore = name.startswith('lay_out') and cases % 2 == 0
        try:
            out = dev.apply(lay, lut, shape, minscore=minscore, restore_order=restore)
            if cases % 3 == 0:                             # the all-device solve must give the same LUT
                lut2, shape2 = dev.solve_lut(t, minscore=minscore)
                check('solve_lut ' + name, shape2 == shape and torch.equal(lut2, lut), info)
        except dev.N.LutNeedsCheckedApply:
            continue
        if getattr(lay, 'seg', None) is not None and not restore:
            out = dev.ungroup(lay, out)
        if isinstance(lay, dev.PairBatch):
            out = lay.unpack(out, b.pitch)
        got_q = out[:n].cpu().numpy().astype(np.int32)
        inside = np.arange(b.pitch)[None, :] < lens[:, None]
        check('qualities ' + name, np.array_equal(np.where(inside, got_q - 33, 0)[:, :S], np.where(inside[:, :S], want_q[:, :S], 0)) and not got_q[~inside].any(), info)
    # length bands (sorted copy of the batch: the only order the reference accepts), each at its own pitch
    if not uniform and n >= 2:
        order = np.argsort(lens, kind='stable')
        sl = lens[order]
        tb = dev.Tables(nrg, 2 * S)
        ok = True
        items = []
        for blo, bhi, longest, shortest in fastx.length_bands(sl):
            pitch = fastx.pitch_for(longest)
            idx = order[blo:bhi]
            band = dev.ReadBatch.from_host(np.ascontiguousarray(host[0][idx, :pitch]), np.ascontiguousarray(host[2][idx, :pitch]),
                                           meta[idx], cseq=np.ascontiguousarray(host[1][idx, :pitch]))
            for lay in (band, dev.group_by_rg(band, nrg)):
                part = dev.Tables(nrg, 2 * S)
                try:
                    dev.accumulate(lay, part, minscore, s_band=longest, s_min=shortest)
                except dev.N.LutNeedsCheckedApply:
                    continue
                ref = dev.Tables(nrg, 2 * S); dev.accumulate(band, ref, minscore)
                check('band %d..%d %s' % (shortest, longest, 'grouped' if lay is not band else 'rows'), torch.equal(part.buf, ref.buf), info)
            dev.accumulate(band, tb, minscore, s_band=longest, s_min=shortest)
            st = dev.meta_stats(band)
            items.append((dev.lay_out(band, nrg, longest, packed=longest <= dev.PACKED_READS and cases % 5 != 0, pairs=False, stats=st), longest, shortest))
        for k, (got, w) in enumerate(zip(tb.to_host(), want[5:9])):
            check('banded tables[%d]' % k, np.array_equal(got, w), info)
        # all bands in ONE launch per kernel (kbbq_accumulate_bands_dev / kbbq_apply_bands_dev) against the oracle's tables and the
        # per-band apply of the same layouts
        try:
            tm = dev.Tables(nrg, 2 * S)
            dev.accumulate_bands(items, tm, minscore)
            for k, (got, w) in enumerate(zip(tm.to_host(), want[5:9])):
                check('merged bands tables[%d]' % k, np.array_equal(got, w), info)
            lutm, shapem = dev.solve_lut(tm, minscore=minscore)
            restore = cases % 2 == 0
            got_o = dev.apply_bands(items, lutm, shapem, minscore=minscore, restore_order=restore)
            for (lay, _, _), g in zip(items, got_o):
                check('merged bands apply', torch.equal(g[:lay.n], dev.apply(lay, lutm, shapem, minscore=minscore, restore_order=restore)[:lay.n]), info)
        except dev.N.LutNeedsCheckedApply:
            pass                                       # a band whose shape the table-driven kernels do not serve
    if cases % 50 == 0:
        print('%d cases, %d mismatches, %.0f s left' % (cases, bad, t_end - time.time()), flush=True)
print('done: %d cases, %d mismatches' % (cases, bad))
sys.exit(1 if bad else 0)
