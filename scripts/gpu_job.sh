# One runner for the GPU box (replaces the per-experiment gpu_r2?.sh / gpu_r3?.sh scripts of rounds 2-3).
# usage, through gpurun:   bash scripts/gpu_job.sh TAG step [step ...]
# Every step writes gpurun_out/<step>_TAG.* and prints one summary line; steps are joined with && semantics (a failed or
# timed-out step ends the job: no further GPU step is started after a kill).
#   tests[=EXPR]      pytest -m gpu (optionally -k EXPR)
#   file=PATH[::K]    pytest -m gpu of one test file (optionally -k K)
#   ranks             the 4-rank rehearsals (tests/test_gpu_ranks.py)
#   bench[=ARGS]      python bench.py (default: the driver's form, --steps 20 --warmup 5)
#   benchq            python bench.py --no-extra --cpu-sample 0 --steps 10 --warmup 3
#   hostentry         scripts/time_host_entry.py (kbbq_accumulate / kbbq_apply on host buffers, PCIe included)
#   e2e[=MODES]       tests/tools/e2e_cli.py: the whole command line on 8 M synthetic reads; MODES e.g. resident,budget=256M,sequential,pipes
#   fuzz=SECONDS      the three randomised campaigns, SECONDS each
#   py=SCRIPT[::ARGS] python SCRIPT ARGS  (timing scripts under scripts/ and tests/tools/)
#   prof=NAME::CMD    rocprofv3 --kernel-trace --stats of CMD (a python command line), summary copied to gpurun_out/prof_NAME_TAG/
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
    ranks)     run ranks 900 python -m pytest tests/test_gpu_ranks.py -x -q -m gpu ;;
    bench)     run bench 900 python bench.py ${arg:---steps 20 --warmup 5}; grep '^{' gpurun_out/bench_$TAG.log | tail -1 > gpurun_out/bench_$TAG.json ;;
    benchq)    run benchq 600 python bench.py --no-extra --cpu-sample 0 --steps 10 --warmup 3; grep '^{' gpurun_out/benchq_$TAG.log | tail -1 > gpurun_out/benchq_$TAG.json ;;
    hostentry) run hostentry 400 python scripts/time_host_entry.py ;;
    e2e)       run e2e 900 python tests/tools/e2e_cli.py --reads 8000000 --reps 3 --modes ${arg:-resident} ;;
    fuzz)      seed=$(( $(date +%s) % 100000 ))
               run fuzz_kernels $((arg + 240)) python tests/tools/fuzz_gpu.py --seconds $arg --seed $seed && run fuzz_aligned $((arg + 240)) python tests/tools/fuzz_gpu_aligned.py --seconds $arg --seed $seed && run fuzz_cli $((arg + 240)) python tests/tools/fuzz_gpu_cli.py --seconds $arg --seed $seed ;;
    py)        s=${arg%%::*}; a=""; [[ "$arg" == *::* ]] && a=${arg#*::}
               run py_$(basename $s .py) 900 python $s $a ;;
    prof)      n=${arg%%::*}; c=${arg#*::}
               rm -rf gpurun_out/prof_${n}_$TAG; run prof_$n 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${n}_$TAG -o $n -- python3 $c
               find gpurun_out/prof_${n}_$TAG -name '*.db' -delete 2> /dev/null ;;
    *)         echo "unknown step $step"; false ;;
  esac || { echo "job stopped at $step"; exit 1; }
done
echo "job $TAG done"
