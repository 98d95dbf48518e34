#!/bin/bash
# usage (GPU box): bash scripts/gpu_pmc_aligned.sh [time_benchmark_path.py arguments] -- PMC counters of K4 / K5 / K6
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_aligned
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TCC_WRITE_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_sum" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $SET -d $OUT/p$i -o p$i --output-format csv -- python $R/scripts/time_benchmark_path.py "$@" > $OUT/p$i.log 2>&1
  echo "pass $i ($SET) rc=$?"
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('$OUT/p*/p*_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if k.startswith(('k4', 'k5', 'k6', 'k1v3')):
            agg[r['Counter_Name']][k].append(float(r['Counter_Value']))
names = sorted({k for c in agg.values() for k in c})
print('| counter | ' + ' | '.join(names) + ' |'); print('|---|' + '---|' * len(names))
for c in agg:
    print('| %s | ' % c + ' | '.join('%.4g' % (sum(agg[c][k]) / len(agg[c][k])) if agg[c][k] else '' for k in names) + ' |')
PY
