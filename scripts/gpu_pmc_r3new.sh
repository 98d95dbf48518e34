# usage (GPU box): bash scripts/gpu_pmc_r3new.sh TAG -- PMC passes (each counter set in its own --pmc run) over round 3's new kernels
TAG=${1:-r03new}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $SET -d $OUT/p$i -o p$i --output-format csv -- python $R/scripts/prof_round3.py > $OUT/p$i.log 2>&1
  rc=$?; echo "pass $i rc=$rc"; [ $rc -eq 0 ] || { tail -5 $OUT/p$i.log; exit $rc; }
done
python - <<PY
import collections, csv, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
want = ('k1v3_bands', 'k2t_bands', 'k1v3_aligned', 'k6_canonical_reads<true>', 'k4v2_find_errors', 'k1v3_accumulate<false, 16, true, 0>', 'k1v3_accumulate<false, 8, true, 0>', 'k2t_apply')
for f in sorted(glob.glob('$OUT/p*/p*_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if any(k.startswith(w) for w in want):
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
names = sorted(agg); ctrs = sorted({c for k in agg for c in agg[k]})
with open('$OUT/table.md', 'w') as fh:
    fh.write('| counter (mean per launch) | ' + ' | '.join(names) + ' |\n|---|' + '---|' * len(names) + '\n')
    fh.write('| launches seen | ' + ' | '.join(str(max(len(v) for v in agg[k].values())) for k in names) + ' |\n')
    for c in ctrs:
        fh.write('| %s | ' % c + ' | '.join('%.4g' % (sum(agg[k][c]) / len(agg[k][c])) if agg[k][c] else '' for k in names) + ' |\n')
print(open('$OUT/table.md').read())
PY
