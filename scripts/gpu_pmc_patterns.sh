#!/bin/bash
# usage (GPU box): bash scripts/gpu_pmc_patterns.sh -- memory-side PMC counters of the copy traversals (scripts/micro/pattern_pmc.hip)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_patterns
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -o $OUT/pattern_pmc $R/scripts/micro/pattern_pmc.hip || exit 1
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_TAG_STALL_sum TCC_BUBBLE_sum TCC_BUSY_sum TCC_CYCLE_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCP_LATENCY_sum" \
           "TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_WRITE_sum TCC_READ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_LATENCY_FIFO_FULL_sum" \
           "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $SET -d $OUT/p$i -o p$i --output-format csv -- $OUT/pattern_pmc > $OUT/p$i.log 2>&1
  echo "pass $i ($SET) rc=$?"
done
rm -f $OUT/pattern_pmc
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('$OUT/p*/p*_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        agg[r['Counter_Name']][k].append(float(r['Counter_Value']))
names = ['tile1k', 'blocked', 'readonly', 'writeonly']
print('| counter | ' + ' | '.join(names) + ' |'); print('|---|' + '---|' * len(names))
for c in agg:
    print('| %s | ' % c + ' | '.join('%.4g' % (sum(agg[c][k]) / len(agg[c][k])) if agg[c][k] else '' for k in names) + ' |')
PY
