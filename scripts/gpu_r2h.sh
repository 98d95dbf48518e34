# usage (GPU box): bash scripts/gpu_r2h.sh TAG SECONDS -- the new GPU tests, then the three randomised campaigns against the CPU oracle
TAG=${1:-r2h}
SEC=${2:-180}
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_pairs.py -x -q -m gpu -k "shorter_band" > gpurun_out/gputests_$TAG.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/gputests_$TAG.log
timeout -k 10 $((SEC + 120)) python tests/tools/fuzz_gpu.py --seconds $SEC --seed 21 > gpurun_out/fuzz_gpu_$TAG.log 2>&1; echo "fuzz_gpu rc=$?"; tail -2 gpurun_out/fuzz_gpu_$TAG.log
timeout -k 10 $((SEC + 120)) python tests/tools/fuzz_gpu_cli.py --seconds $SEC --seed 22 > gpurun_out/fuzz_cli_$TAG.log 2>&1; echo "fuzz_gpu_cli rc=$?"; tail -2 gpurun_out/fuzz_cli_$TAG.log
timeout -k 10 $((SEC + 120)) python tests/tools/fuzz_gpu_aligned.py --seconds $SEC --seed 23 > gpurun_out/fuzz_aligned_$TAG.log 2>&1; echo "fuzz_gpu_aligned rc=$?"; tail -2 gpurun_out/fuzz_aligned_$TAG.log
