#!/usr/bin/env python3
"""profiles/pmc_traffic.json from FETCH_SIZE / WRITE_SIZE passes (rocprofv3 --pmc, one counter per pass, scripts/gpu_job.sh pmc):
HBM bytes per base of K1 / K2 for every device layout, keyed to the kernel source they were taken on (bench.py prints
`traffic` only when its own kernel_source_sha matches)."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
root, reads, readlen = sys.argv[1], int(sys.argv[2]), 150
bases = reads * readlen
out = {'_comment': 'HBM bytes per base from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, scripts/gpu_job.sh pmc), %d reads x %d bp '
                   'per launch; FETCH_SIZE is doubled (gfx950 tallies the 128-byte requests of coalesced streaming reads at 64 bytes, '
                   'MI355X_MICROARCH.md HBM section); both counters are in KiB.  Layout keys: bench.py layout_key().' % (reads, readlen),
       'kernel_source_sha': bench.kernel_source_sha(), 'bases_per_launch_measured': bases}
for layout in sorted(os.listdir(root)):
    d = os.path.join(root, layout)
    if not os.path.isdir(d):
        continue
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + '/p*/p*_counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].replace('void ', '')
            name = 'k1_accumulate' if k.startswith('k1v3_accumulate') or k.startswith('k1_accumulate') else 'k2_apply' if k.startswith(('k2v3_apply', 'k2_apply', 'k2t_apply')) else None
            if name and r['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'):
                vals[name][r['Counter_Name']].append(float(r['Counter_Value']))
    entry = {}
    for name, c in vals.items():
        if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
            fetch, write = sum(c['FETCH_SIZE']) / len(c['FETCH_SIZE']), sum(c['WRITE_SIZE']) / len(c['WRITE_SIZE'])
            entry[name] = {'FETCH_SIZE_KB': fetch, 'WRITE_SIZE_KB': write, 'hbm_bytes_per_base': round((2 * fetch + write) * 1024 / bases, 4)}
    if entry:
        out[layout] = entry
json.dump(out, sys.stdout, indent=1)
print()
