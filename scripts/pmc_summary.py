#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSVs (one counter set per run) into a per-kernel table (markdown)."""
import collections, csv, glob, sys
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(d + '/p*/p*_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if not (k.startswith('k1') or k.startswith('k2') or k.startswith('k3')):
            continue
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
names = sorted(agg)
ctrs = sorted({c for k in agg for c in agg[k]})
print('| counter | ' + ' | '.join(names) + ' |')
print('|---|' + '---|' * len(names))
for c in ctrs:
    print('| %s | ' % c + ' | '.join('%.4g' % (sum(agg[k][c]) / len(agg[k][c])) if agg[k][c] else '' for k in names) + ' |')
