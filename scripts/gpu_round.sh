set -x
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -q -m gpu > gpurun_out/gputests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/gputests.log
timeout -k 10 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench1.json 2> gpurun_out/bench1.err; echo "bench rc=$?"; cat gpurun_out/bench1.json; tail -5 gpurun_out/bench1.err
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1 -o r1 --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof_bench.err; echo "rocprof rc=$?"
cat $R/gpurun_out/prof_bench.json
ls -R $R/gpurun_out/prof_r1 | head -20
