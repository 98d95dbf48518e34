# usage (GPU box): bash scripts/gpu_prof_bench.sh TAG [bench args] -- bench.py as the driver runs it, then the same command under rocprofv3 --kernel-trace --stats
TAG=${1:-prof}; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/stats_$TAG
mkdir -p $OUT
cd $R
timeout -k 10 500 python bench.py "$@" > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $OUT -o s --output-format csv -- python $R/bench.py "$@" > $R/gpurun_out/bench_prof_$TAG.json 2> $R/gpurun_out/bench_prof_$TAG.err; echo "rocprof rc=$?"
python - <<PY
import csv, glob
for f in glob.glob('$OUT/**/s_kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r['TotalDurationNs']))
    for r in rows[:24]:
        print('%-64s calls %5s avg %10.1f us  min %10.1f  max %10.1f' % (r['Name'][:64], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
