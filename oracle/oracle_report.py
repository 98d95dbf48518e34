"""
oracle_report.py -- CPU restatement of the GATK-report model codec.
TEST INFRASTRUCTURE ONLY: imported by tests/ (and gen_golden.py); the product never uses it.

Plain Python loops and lists, no pandas: an independent second statement of
  * kbbq/gatk/bqsr.py:226-366  vectors_to_report   -> report_text()
  * kbbq/recaltable.py:186-347 table text format    -> render_table() / parse_report()
  * kbbq/gatk/applybqsr.py:14-44 table_to_vectors  -> table_to_vectors()
Pinned by (a) the reference's own known answers (tests/test_recaltable.py:75-83 table
string, tests/test_gatk_applybqsr.py:13-63 small report -> vectors), restated as data in
tests/test_oracle_report.py, and (b) reports written by the UNMODIFIED reference
vectors_to_report for the golden count vectors (oracle/gen_golden.py -> tests/golden/
report_*.json|txt).  The reference's table_to_vectors does not run on the pandas installed
here (.loc with absent labels is a KeyError since pandas 1.0); it is pinned by (a) and by the
round trip through the reference-written reports.
"""
import numpy as np

import oracle as O

ARGUMENTS = [  # bqsr.py:263-281
    ('binary_tag_name', 'null'),
    ('covariate', 'ReadGroupCovariate,QualityScoreCovariate,ContextCovariate,CycleCovariate'),
    ('default_platform', 'null'), ('deletions_default_quality', '45'), ('force_platform', 'null'),
    ('indels_context_size', '3'), ('insertions_default_quality', '45'), ('low_quality_tail', '2'),
    ('maximum_cycle_value', '500'), ('mismatches_context_size', '2'), ('mismatches_default_quality', '-1'),
    ('no_standard_covs', 'false'), ('quantizing_levels', '16'), ('recalibration_report', 'null'),
    ('run_without_dbsnp', 'false'), ('solid_nocall_strategy', 'THROW_EXCEPTION'),
    ('solid_recal_mode', 'SET_Q_ZERO')]
DINUCS = [a + b for a in 'ATGC' for b in 'ATGC']          # compare_reads.py:199-214


def render_table(title, description, header, fmts, rows):
    """recaltable.py:246-347.  rows: list of tuples.  Widths: strings by their own length;
    numeric columns by the value rendered with the LAST column's format (the reference's
    formatter closures all bind the last format, :330), never narrower than the header."""
    widths = [len(h) for h in header]
    if rows:
        for i, f in enumerate(fmts):
            if f == '%s':
                w = max(len(r[i]) for r in rows)
            else:
                w = max(len(fmts[-1] % r[i]) for r in rows)
            widths[i] = max(widths[i], w)
    out = [':'.join(['#', 'GATKTable', str(len(header)), str(len(rows))] + fmts + [';']),
           ':'.join(['#', 'GATKTable', title, description]),
           '  '.join(h.ljust(w) for h, w in zip(header, widths))]
    for r in rows:
        out.append('  '.join(r[i].ljust(widths[i]) if f == '%s' else (f % float(r[i])).rjust(widths[i])
                             for i, f in enumerate(fmts)))
    return '\n'.join(out)


def report_text(meanq, rg_errs, rg_total, q_errs, q_total, pos_errs, pos_total,
                dinuc_errs, dinuc_total, rg_order):
    """bqsr.py:226-366 followed by str(RecalibrationReport) (recaltable.py:99-106,479-491)."""
    R, Q = q_total.shape
    tables = [render_table('Arguments', 'Recalibration argument collection values used in this run',
                           ['Argument', 'Value'], ['%s', '%s'], ARGUMENTS)]
    # Quantized (bqsr.py:313-323): counts per score, identity map, unobserved -> 93
    seen = q_total.sum(axis=0)
    qrows = []
    for s in range(94):
        cnt = int(seen[s]) if s < Q else 0
        qrows.append((s, cnt, s if (s < Q and cnt != 0) else 93))
    tables.append(render_table('Quantized', 'Quality quantization map',
                               ['QualityScore', 'Count', 'QuantizedScore'], ['%d', '%d', '%d'], qrows))
    # RecalTable0 (bqsr.py:288-300)
    with np.errstate(divide='ignore', invalid='ignore'):
        est = -10.0 * np.log10(np.sum(O.q_to_p(np.arange(Q)) * q_total, axis=1) / rg_total).round(decimals=5).astype(np.float64)
    est[np.isnan(est)] = 0
    emp = (O.gatk_delta_q(est, rg_errs.copy(), rg_total.copy()) + est).astype(np.float64)
    rows = [(str(rg_order[r]), 'M', emp[r], est[r], int(rg_total[r]), float(rg_errs[r]))
            for r in range(R) if rg_total[r] != 0]
    tables.append(render_table('RecalTable0', '', ['ReadGroup', 'EventType', 'EmpiricalQuality',
                               'EstimatedQReported', 'Observations', 'Errors'],
                               ['%s', '%s', '%.4f', '%.4f', '%d', '%.2f'], rows))
    # RecalTable1 (bqsr.py:302-311)
    qq = np.broadcast_to(np.arange(Q), (R, Q)).copy()
    empq = O.gatk_delta_q(qq, q_errs, q_total) + qq
    rows = [(str(rg_order[r]), q, 'M', float(empq[r, q]), int(q_total[r, q]), float(q_errs[r, q]))
            for r in range(R) for q in range(Q) if q_total[r, q] != 0]
    tables.append(render_table('RecalTable1', '', ['ReadGroup', 'QualityScore', 'EventType',
                               'EmpiricalQuality', 'Observations', 'Errors'],
                               ['%s', '%d', '%s', '%.4f', '%d', '%.2f'], rows))
    # RecalTable2 (bqsr.py:325-358): Context + Cycle rows, sorted by (RG, Q, value text, name)
    S2 = pos_total.shape[2]
    n = S2 // 2
    cyc = [str(c + 1) for c in range(n)] + [str(-(n - c)) for c in range(n)]
    pq = np.broadcast_to(np.arange(Q)[None, :, None], pos_total.shape).copy()
    dq = np.broadcast_to(np.arange(Q)[None, :, None], dinuc_total.shape).copy()
    emp_pos = O.gatk_delta_q(pq, pos_errs, pos_total) + pq
    emp_dn = O.gatk_delta_q(dq, dinuc_errs, dinuc_total) + dq
    rows = []
    for r in range(R):
        for q in range(Q):
            for d in range(16):
                if dinuc_total[r, q, d] != 0:
                    rows.append((str(rg_order[r]), q, DINUCS[d], 'Context', 'M', float(emp_dn[r, q, d]),
                                 int(dinuc_total[r, q, d]), float(dinuc_errs[r, q, d])))
            for c in range(S2):
                if pos_total[r, q, c] != 0:
                    rows.append((str(rg_order[r]), q, cyc[c], 'Cycle', 'M', float(emp_pos[r, q, c]),
                                 int(pos_total[r, q, c]), float(pos_errs[r, q, c])))
    rows.sort(key=lambda t: (t[0], t[1], t[2], t[3]))
    tables.append(render_table('RecalTable2', '', ['ReadGroup', 'QualityScore', 'CovariateValue',
                               'CovariateName', 'EventType', 'EmpiricalQuality', 'Observations', 'Errors'],
                               ['%s', '%d', '%s', '%s', '%s', '%.4f', '%d', '%.2f'], rows))
    return '#:GATKReport.v1.1:5\n' + ''.join(t + '\n\n' for t in tables)


def parse_report(text):
    """recaltable.py:46-66,186-244: [(title, description, header, fmts, rows of str)]."""
    first, _, body = text.partition('\n')
    _, version, ntables = first.strip().split(':')
    out = []
    for chunk in body.split('\n\n'):
        if chunk == '':
            continue
        lines = chunk.splitlines()
        fmts = lines[0].split(':')[4:-1]
        title, description = lines[1].split(':')[2:4]
        out.append((title, description, lines[2].split(), fmts, [ln.split() for ln in lines[3:]]))
    if len(out) != int(ntables):
        raise ValueError('truncated report')
    return out


def table_to_vectors(text, rg_order, maxscore=42):
    """applybqsr.py:14-44 on the report text."""
    tabs = {t[0]: t for t in parse_report(text)}
    rgi = {str(n): i for i, n in enumerate(rg_order)}
    R, Q = len(rg_order), maxscore + 1
    meanq = np.zeros(R); ge = np.zeros(R, dtype=np.int64); gt = np.zeros(R, dtype=np.int64)
    for rg, _, _, estq, obs, errs in tabs['RecalTable0'][4]:
        if rg in rgi:
            meanq[rgi[rg]], gt[rgi[rg]], ge[rgi[rg]] = float(estq), int(obs), int(float(errs))
    qe = np.zeros((R, Q), dtype=np.int64); qt = np.zeros((R, Q), dtype=np.int64)
    for rg, q, _, _, obs, errs in tabs['RecalTable1'][4]:
        if rg in rgi and 0 <= int(q) < Q:
            qt[rgi[rg], int(q)], qe[rgi[rg], int(q)] = int(obs), int(float(errs))
    rows = [r for r in tabs['RecalTable2'][4] if r[0] in rgi and 0 <= int(r[1]) < Q]
    seqlen = max(int(r[2]) for r in rows if r[3] == 'Cycle')
    pe = np.zeros((R, Q, 2 * seqlen), dtype=np.int64); pt = np.zeros_like(pe)
    de = np.zeros((R, Q, 16), dtype=np.int64); dt = np.zeros_like(de)
    for rg, q, value, name, _, _, obs, errs in rows:
        if name == 'Cycle':
            c = int(value)
            col = c - 1 if c > 0 else 2 * seqlen + c
            if c != 0 and c >= -seqlen:
                pt[rgi[rg], int(q), col], pe[rgi[rg], int(q), col] = int(obs), int(float(errs))
        elif name == 'Context' and value in DINUCS:
            dt[rgi[rg], int(q), DINUCS.index(value)], de[rgi[rg], int(q), DINUCS.index(value)] = int(obs), int(float(errs))
    return meanq, ge, gt, qe, qt, pe, pt, de, dt
