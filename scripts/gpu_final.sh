# usage (GPU box): bash scripts/gpu_final.sh TAG -- tests, bench (with CPU baseline), rocprofv3 kernel stats of the same command
TAG=${1:-final}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 600 python -m pytest tests -q -m gpu > gpurun_out/gputests_$TAG.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/gputests_$TAG.log
timeout -k 10 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"; cat gpurun_out/bench_$TAG.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/stats_$TAG -o s --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --cpu-sample 0 > $R/gpurun_out/bench_prof_$TAG.json 2> $R/gpurun_out/bench_prof_$TAG.err; echo "rocprof rc=$?"
cat $R/gpurun_out/bench_prof_$TAG.json
head -6 $R/gpurun_out/stats_$TAG/s_kernel_stats.csv
