# usage (GPU box): bash scripts/gpu_rowlut.sh -- K2 on one-read-per-row planes with the LUT narrowed to the rows' pitch: tests, A/B, mixed lengths
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_layouts.py tests/test_gpu_pairs.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/gputests_rowlut.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/gputests_rowlut.log
for ROUND in 1 2; do
echo "--- one read per row, 4-bit planes, 50 M x 150: default / KBBQ_K2_TILE=0 / KBBQ_K2_ROWLUT=0 (tile off too)"
timeout -k 10 200 python scripts/time_kernels.py --reads 50000000 --packed --single 2>&1 | tail -1
KBBQ_K2_TILE=0 timeout -k 10 200 python scripts/time_kernels.py --reads 50000000 --packed --single 2>&1 | tail -1
KBBQ_K2_ROWLUT=0 KBBQ_K2_TILE=0 timeout -k 10 200 python scripts/time_kernels.py --reads 50000000 --packed --single 2>&1 | tail -1
echo "--- one read per row, characters, 50 M x 150: default / KBBQ_K2_ROWLUT=0"
timeout -k 10 200 python scripts/time_kernels.py --reads 50000000 2>&1 | tail -1
KBBQ_K2_ROWLUT=0 timeout -k 10 200 python scripts/time_kernels.py --reads 50000000 2>&1 | tail -1
echo "--- mixed lengths: default / KBBQ_K2_TILE=0 / KBBQ_K2_ROWLUT=0"
timeout -k 10 300 python scripts/time_mixed.py 2>&1 | tail -1 | cut -c1-900
KBBQ_K2_TILE=0 timeout -k 10 300 python scripts/time_mixed.py 2>&1 | tail -1 | cut -c1-900
KBBQ_K2_ROWLUT=0 timeout -k 10 300 python scripts/time_mixed.py 2>&1 | tail -1 | cut -c1-900
done
