# usage (GPU box): bash scripts/gpu_r2e.sh TAG -- K1 / K2 layout tests, then kernel timings of the packed layout (A/B switches in the environment)
TAG=${1:-r2e}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_layouts.py tests/test_gpu_pairs.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/gputests_$TAG.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/gputests_$TAG.log
echo "--- packed, default"
timeout -k 10 200 python scripts/time_kernels.py --reads 50000000 --packed 2>&1 | tail -1
echo "--- packed, KBBQ_K2_KM=0 (position-major pair LUT)"
KBBQ_K2_KM=0 timeout -k 10 200 python scripts/time_kernels.py --reads 50000000 --packed 2>&1 | tail -1
echo "--- packed, KBBQ_K1_KM=0 (position-major cycle table)"
KBBQ_K1_KM=0 timeout -k 10 200 python scripts/time_kernels.py --reads 50000000 --packed 2>&1 | tail -1
echo "--- pairs on character planes, default / KBBQ_K2_KM=0"
timeout -k 10 200 python scripts/time_kernels.py --reads 50000000 --pairs 2>&1 | tail -1
KBBQ_K2_KM=0 timeout -k 10 200 python scripts/time_kernels.py --reads 50000000 --pairs 2>&1 | tail -1
echo "--- packed, 8 read groups"
timeout -k 10 200 python scripts/time_kernels.py --reads 50000000 --rgs 8 --packed 2>&1 | tail -1
