# usage (GPU box): bash scripts/gpu_pmc_r3.sh TAG -- round-3 PMC passes (each counter set in its own run, --pmc only): FETCH_SIZE / WRITE_SIZE of
# K1 / K2 on every layout (profiles/pmc_traffic.json, keyed to the kernel sources) and the counter table of K1 / K2 / K3 on the bench layout
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT/table $OUT/traffic/pairs_nib $OUT/traffic/pairs $OUT/traffic/reads
cd /tmp && export TMPDIR=/tmp
for LAY in pairs_nib pairs reads; do
  case $LAY in pairs_nib) ARG="--packed";; pairs) ARG="--pairs";; reads) ARG="";; esac
  i=0
  for SET in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $SET -d $OUT/traffic/$LAY/p$i -o p$i --output-format csv -- python $R/scripts/prof_kernels.py --reads 20000000 --reps 2 $ARG > $OUT/traffic/$LAY/p$i.log 2>&1
    rc=$?; echo "traffic $LAY pass $i ($SET) rc=$rc"
    [ $rc -eq 0 ] || exit $rc
  done
done
python $R/scripts/pmc_traffic_json.py $OUT/traffic 20000000 > $OUT/pmc_traffic.json; cat $OUT/pmc_traffic.json
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $SET -d $OUT/table/p$i -o p$i --output-format csv -- python $R/scripts/prof_kernels.py --reads 20000000 --reps 2 --packed > $OUT/table/p$i.log 2>&1
  rc=$?; echo "table pass $i rc=$rc"
  [ $rc -eq 0 ] || exit $rc
done
python $R/scripts/pmc_summary.py $OUT/table > $OUT/table.md; cat $OUT/table.md
