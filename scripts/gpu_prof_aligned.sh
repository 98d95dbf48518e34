# usage (GPU box): bash scripts/gpu_prof_aligned.sh TAG [args of time_benchmark_path.py] -- rocprofv3 kernel stats of the aligned-read kernels
TAG=${1:-aligned}; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/stats_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT -o s --output-format csv -- python $R/scripts/time_benchmark_path.py "$@" > $OUT/run.log 2>&1; echo "rocprof rc=$?"
grep "^K4\|^K5\|^K6" $OUT/run.log
python - <<PY
import csv, glob
for f in glob.glob('$OUT/**/s_kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Name'].startswith(('k4', 'k5', 'k6', 'void k4', 'void k5', 'void k6', 'void k1v3')) or 'k4' in r['Name']:
            print('%-60s calls %4s avg %10.1f us  min %10.1f  max %10.1f' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
