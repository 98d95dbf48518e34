# One runner for the GPU box (replaces the per-experiment gpu_r2?.sh / gpu_r3?.sh scripts of rounds 2-3).
# usage, through gpurun:   bash scripts/gpu_job.sh TAG step [step ...]
# Every step writes gpurun_out/<step>_TAG.* and prints one summary line; steps are joined with && semantics (a failed or
# timed-out step ends the job: no further GPU step is started after a kill).
#   tests[=EXPR]      pytest -m gpu (optionally -k EXPR)
#   file=PATH[::K]    pytest -m gpu of one test file (optionally -k K)
#   ranks[=N]         the N-rank rehearsals (tests/test_gpu_ranks.py; default 4: all the box's process guard admits)
#   bench[=ARGS]      python bench.py (default: the driver's form, --steps 20 --warmup 5)
#   benchq            python bench.py --no-extra --cpu-sample 0 --steps 10 --warmup 3
#   hostentry         scripts/time_host_entry.py (kbbq_accumulate / kbbq_apply on host buffers, PCIe included)
#   smoke             __graft_entry__.smoke()
#   e2e[=MODES]       tests/tools/e2e_cli.py: the whole command line on 8 M synthetic reads; MODES e.g. resident,budget=256M,sequential,pipes
#   fuzz=SECONDS      the three randomised campaigns, SECONDS each
#   py=SCRIPT[::ARGS] python SCRIPT ARGS  (timing scripts under scripts/ and tests/tools/)
#   prof=NAME::CMD    rocprofv3 --kernel-trace --stats of CMD (a python command line), summary copied to gpurun_out/prof_NAME_TAG/
#   pmc[=table]       FETCH_SIZE / WRITE_SIZE passes of K1 / K2 on every layout -> profiles/pmc_traffic.json (+ the counter table of the bench layout)
#   clitrace          which kernels `kbbq recalibrate -f` launches (tests/tools/trace_cli.sh: rocprofv3 kernel trace of the command line)
#   evidence          bench.py as the driver runs it + rocprofv3 --kernel-trace --stats of the same command (with and without `extra`)
TAG=$1; shift
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
run() {  # run NAME TIMEOUT CMD...
  local name=$1 limit=$2; shift 2
  local log=gpurun_out/${name}_$TAG.log
  echo "== $name: $*" > $log
  timeout -k 10 $limit "$@" >> $log 2>&1
  local rc=$?
  echo "$name rc=$rc: $(tail -1 $log | cut -c1-300)"
  return $rc
}
for step in "$@"; do
  kind=${step%%=*}; arg=""; [[ "$step" == *=* ]] && arg=${step#*=}
  case $kind in
    tests)     if [ -n "$arg" ]; then run tests 1100 python -m pytest tests -x -q -m gpu -k "$arg"; else run tests 1100 python -m pytest tests -x -q -m gpu; fi ;;
    file)      f=${arg%%::*}; k=""; [[ "$arg" == *::* ]] && k=${arg#*::}
               if [ -n "$k" ]; then run file_$(basename $f .py) 1000 python -m pytest $f -x -q -m gpu -k "$k"; else run file_$(basename $f .py) 1000 python -m pytest $f -x -q -m gpu; fi ;;
    ranks)     KBBQ_TEST_RANKS=${arg:-4} run ranks 900 python -m pytest tests/test_gpu_ranks.py -x -q -m gpu ;;
    bench)     run bench 900 python bench.py ${arg:---steps 20 --warmup 5}; grep '^{' gpurun_out/bench_$TAG.log | tail -1 > gpurun_out/bench_$TAG.json ;;
    benchq)    run benchq 600 python bench.py --no-extra --cpu-sample 0 --steps 10 --warmup 3; grep '^{' gpurun_out/benchq_$TAG.log | tail -1 > gpurun_out/benchq_$TAG.json ;;
    hostentry) run hostentry 400 python scripts/time_host_entry.py ;;
    smoke)     run smoke 600 python -c 'import __graft_entry__ as g; g.smoke()' ;;
    e2e)       run e2e 900 python tests/tools/e2e_cli.py --reads ${E2E_READS:-8000000} --reps ${E2E_REPS:-3} --modes ${arg:-resident} ;;
    fuzz)      seed=$(( $(date +%s) % 100000 ))
               run fuzz_kernels $((arg + 240)) python tests/tools/fuzz_gpu.py --seconds $arg --seed $seed && run fuzz_aligned $((arg + 240)) python tests/tools/fuzz_gpu_aligned.py --seconds $arg --seed $seed && run fuzz_cli $((arg + 240)) python tests/tools/fuzz_gpu_cli.py --seconds $arg --seed $seed ;;
    py)        s=${arg%%::*}; a=""; [[ "$arg" == *::* ]] && a=${arg#*::}
               run py_$(basename $s .py) 900 python $s $a ;;
    prof)      n=${arg%%::*}; c=${arg#*::}
               rm -rf gpurun_out/prof_${n}_$TAG; run prof_$n 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${n}_$TAG -o $n -- python3 $c
               find gpurun_out/prof_${n}_$TAG -name '*.db' -delete 2> /dev/null ;;
    pmc)       # HBM traffic of K1 / K2 on THIS tree's kernel sources (FETCH_SIZE / WRITE_SIZE, each counter in its own --pmc run; arg "table": the
               # counter table of the bench layout too).  profiles/pmc_traffic.json is what bench.py's roofline.traffic reads.
               OUT=$PWD/gpurun_out/pmc_$TAG; mkdir -p $OUT; ok=1
               for LAY in pairs_nib pairs reads; do
                 case $LAY in pairs_nib) A="--packed";; pairs) A="--pairs";; reads) A="";; esac
                 i=0
                 for SET in FETCH_SIZE WRITE_SIZE; do
                   i=$((i+1)); mkdir -p $OUT/traffic/$LAY
                   timeout -k 10 300 rocprofv3 --pmc $SET -d $OUT/traffic/$LAY/p$i -o p$i --output-format csv -- python3 $PWD/scripts/prof_kernels.py --reads 20000000 --reps 2 $A > $OUT/traffic/$LAY/p$i.log 2>&1 || ok=0
                 done
               done
               [ $ok -eq 1 ] && python scripts/pmc_traffic_json.py $OUT/traffic 20000000 > $OUT/pmc_traffic.json && cp $OUT/pmc_traffic.json profiles/pmc_traffic.json \
                 && cp $OUT/pmc_traffic.json gpurun_out/pmc_traffic_$TAG.json && echo "pmc traffic: $(grep -o '"hbm_bytes_per_base": [0-9.]*' $OUT/pmc_traffic.json | tr '\n' ' ')"
               if [ $ok -eq 1 ] && [ "$arg" = "table" ]; then
                 i=0
                 for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
                            "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
                            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU" \
                            "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
                   i=$((i+1)); mkdir -p $OUT/table
                   timeout -k 10 300 rocprofv3 --pmc $SET -d $OUT/table/p$i -o p$i --output-format csv -- python3 $PWD/scripts/prof_kernels.py --reads 20000000 --reps 2 --packed > $OUT/table/p$i.log 2>&1 || ok=0
                 done
                 [ $ok -eq 1 ] && python scripts/pmc_summary.py $OUT/table > gpurun_out/pmc_table_$TAG.md
               fi
               find $OUT -name '*.db' -delete 2> /dev/null
               [ $ok -eq 1 ] ;;
    evidence)  # the round's evidence run: bench.py as the driver runs it, then the same command under rocprofv3 --kernel-trace --stats (headline
               # only, and with the extra object); summaries land in gpurun_out/ for copying into profiles/rNN_*
               run bench 900 python bench.py --steps 20 --warmup 5 && grep '^{' gpurun_out/bench_$TAG.log | tail -1 > gpurun_out/bench_$TAG.json \
               && rm -rf gpurun_out/stats_$TAG gpurun_out/stats_${TAG}_extra \
               && run prof_headline 600 rocprofv3 --kernel-trace --stats -d gpurun_out/stats_$TAG -o s --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-extra --cpu-sample 0 \
               && grep '^{' gpurun_out/prof_headline_$TAG.log | tail -1 > gpurun_out/bench_under_rocprof_$TAG.json \
               && run prof_extra 900 rocprofv3 --kernel-trace --stats -d gpurun_out/stats_${TAG}_extra -o s --output-format csv -- python3 bench.py --steps 20 --warmup 5 --cpu-sample 0 \
               && cp $(find gpurun_out/stats_$TAG -name 's_kernel_stats.csv' | head -1) gpurun_out/bench_kernel_stats_$TAG.csv \
               && cp $(find gpurun_out/stats_$TAG -name 's_kernel_trace.csv' | head -1) gpurun_out/bench_kernel_trace_$TAG.csv \
               && cp $(find gpurun_out/stats_${TAG}_extra -name 's_kernel_stats.csv' | head -1) gpurun_out/bench_with_extra_kernel_stats_$TAG.csv \
               && find gpurun_out/stats_$TAG gpurun_out/stats_${TAG}_extra -name '*.db' -delete 2> /dev/null; head -8 gpurun_out/bench_kernel_stats_$TAG.csv ;;
    clitrace)  run clitrace 600 bash tests/tools/trace_cli.sh $TAG ;;
    *)         echo "unknown step $step"; false ;;
  esac || { echo "job stopped at $step"; exit 1; }
done
echo "job $TAG done"
