# usage (GPU box): bash scripts/gpu_k1two.sh -- K1 with two workgroups per CU (8 context-table copies of 25 words: 79 KB of LDS) against one (16 copies of 32)
# second builds beforehand (the two macros and the workgroups-per-CU computation were a patch of the experiment, not kept in the tree):
#   make -C kbbq-py_amd/csrc OUT=../kbbq/libkbbq_hip_v1.so EXTRA="-DK1V3_DNREP=8 -DK1V3_DNSLOTS=25"   (two workgroups per CU)
#                           make -C kbbq-py_amd/csrc OUT=../kbbq/libkbbq_hip_v2.so EXTRA="-DK1V3_DNREP=8"                      (8 copies, still one workgroup)
R=$GRAFT_REPO_ROOT
cd $R
for V in v1 v2; do
KBBQ_HIP_LIB=$R/kbbq-py_amd/kbbq/libkbbq_hip_$V.so timeout -k 10 900 python -m pytest tests/test_gpu_layouts.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/gputests_k1$V.log 2>&1; echo "pytest ($V) rc=$?"; tail -1 gpurun_out/gputests_k1$V.log
done
for ROUND in 1 2 3; do
for V in base v1 v2; do
if [ $V = base ]; then L=$R/kbbq-py_amd/kbbq/libkbbq_hip.so; else L=$R/kbbq-py_amd/kbbq/libkbbq_hip_$V.so; fi
echo "--- round $ROUND, $V: pairs 4-bit / 8 read groups / mixed"
KBBQ_HIP_LIB=$L timeout -k 10 200 python scripts/time_kernels.py --reads 50000000 --packed 2>&1 | tail -1
KBBQ_HIP_LIB=$L timeout -k 10 200 python scripts/time_kernels.py --reads 50000000 --packed --rgs 8 2>&1 | tail -1
KBBQ_HIP_LIB=$L timeout -k 10 300 python scripts/time_mixed.py 2>&1 | tail -1 | cut -c330-470
done
done
