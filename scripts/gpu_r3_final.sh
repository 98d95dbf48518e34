# usage (GPU box): bash scripts/gpu_r3_final.sh TAG [quick] -- round 3's evidence run: GPU tests, bench.py as the driver runs it, rocprofv3 kernel
# stats of the same command (headline only, and with the `extra` object), the command line's kernel trace and timeline
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
# HBM traffic of K1 / K2 on the kernel sources of THIS tree first (FETCH_SIZE / WRITE_SIZE, each counter in its own --pmc run): bench.py prints
# roofline.traffic only from a profiles/pmc_traffic.json taken on its own sources
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT/traffic/pairs_nib $OUT/traffic/pairs $OUT/traffic/reads
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
python $R/scripts/pmc_traffic_json.py $OUT/traffic 20000000 > $OUT/pmc_traffic.json && cp $OUT/pmc_traffic.json $R/profiles/pmc_traffic.json
grep -o '"hbm_bytes_per_base": [0-9.]*' $OUT/pmc_traffic.json | tr '\n' ' '; echo
cd $R
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/gputests_$TAG.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -6 gpurun_out/gputests_$TAG.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 700 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
rc=$?; echo "bench rc=$rc"; tail -2 gpurun_out/bench_$TAG.err
[ $rc -eq 0 ] || exit $rc
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/stats_$TAG -o s --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-extra --cpu-sample 0 > $R/gpurun_out/bench_prof_$TAG.json 2> $R/gpurun_out/bench_prof_$TAG.err
rc=$?; echo "rocprof (headline) rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/stats_${TAG}_extra -o s --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --cpu-sample 0 > $R/gpurun_out/bench_prof_${TAG}_extra.json 2> $R/gpurun_out/bench_prof_${TAG}_extra.err
rc=$?; echo "rocprof (with extra) rc=$rc"
[ $rc -eq 0 ] || exit $rc
python - <<PY
import csv, glob
for tag in ('$TAG', '${TAG}_extra'):
    for f in glob.glob('$R/gpurun_out/stats_%s/**/s_kernel_stats.csv' % tag, recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r['TotalDurationNs']))
        print('---', tag)
        for r in rows[:14 if tag == '$TAG' else 40]:
            print('%-70s calls %5s avg %10.1f us  min %10.1f  max %10.1f' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
cd $R
bash scripts/gpu_trace_cli.sh $TAG > gpurun_out/trace_cli_$TAG.txt 2>&1
rc=$?; echo "cli trace rc=$rc"; cat gpurun_out/trace_cli_$TAG.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 bash scripts/e2e_timeline.sh 8000000 > gpurun_out/timeline_$TAG.log 2>&1
rc=$?; echo "timeline rc=$rc"; grep -v " write$\| format$\| D2H$" gpurun_out/timeline_$TAG.log
exit $rc
