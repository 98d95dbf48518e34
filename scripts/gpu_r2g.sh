# usage (GPU box): bash scripts/gpu_r2g.sh -- mid-round check: layout / pair tests and kernel timings
TAG=${1:-r2g}
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_layouts.py tests/test_gpu_pairs.py -x -q -m gpu > gpurun_out/gputests_$TAG.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/gputests_$TAG.log
timeout -k 10 200 python scripts/time_layout.py --rgs 1 2>&1 | grep lay_out
timeout -k 10 200 python scripts/time_layout.py --rgs 8 2>&1 | grep lay_out
