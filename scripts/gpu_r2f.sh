# usage (GPU box): bash scripts/gpu_r2f.sh TAG -- whole GPU suite
TAG=${1:-r2f}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/gputests_$TAG.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/gputests_$TAG.log
