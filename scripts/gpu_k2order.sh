# usage (GPU box): bash scripts/gpu_k2order.sh -- K2 storing through the permutation: workgroups of all groups advancing together vs group after group
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_layouts.py tests/test_gpu_pairs.py -x -q -m gpu > gpurun_out/gputests_k2order.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/gputests_k2order.log
for ROUND in 1 2 3; do
for RG in 8 32; do
echo "--- round $ROUND rgs $RG: interleaved / KBBQ_K2_ORDER=0 / not through the permutation"
timeout -k 10 200 python scripts/time_kernels.py --reads 50000000 --packed --rgs $RG --restore 2>&1 | tail -1
KBBQ_K2_ORDER=0 timeout -k 10 200 python scripts/time_kernels.py --reads 50000000 --packed --rgs $RG --restore 2>&1 | tail -1
timeout -k 10 200 python scripts/time_kernels.py --reads 50000000 --packed --rgs $RG 2>&1 | tail -1
done
done
