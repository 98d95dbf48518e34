# usage (GPU box): bash scripts/gpu_one_test.sh PYTEST_ARGS... -- one test selection, log under gpurun_out/
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest "$@" -x -q -m gpu > gpurun_out/gputests_one.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -30 gpurun_out/gputests_one.log; exit $rc
