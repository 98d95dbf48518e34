# usage (GPU box): bash scripts/gpu_k1threads.sh -- K1 with 512- / 768-thread workgroups (second builds) against 1024
# (the second builds: a copy of kbbq-py_amd/csrc with K1V3_THREADS edited, `make OUT=.../kbbq/libkbbq_hip_t512.so` / `_t768.so`, before gpurun;
#  KBBQ_HIP_LIB selects the library a process loads)
R=$GRAFT_REPO_ROOT
cd $R
for T in 512 768; do
KBBQ_HIP_LIB=$R/kbbq-py_amd/kbbq/libkbbq_hip_t$T.so timeout -k 10 600 python -m pytest tests/test_gpu_layouts.py -x -q -m gpu -k "lay_out_tally or ragged" > gpurun_out/gputests_k1t$T.log 2>&1; echo "pytest ($T threads) rc=$?"; tail -1 gpurun_out/gputests_k1t$T.log
done
for ROUND in 1 2 3; do
for T in 1024 768 512; do
if [ $T = 1024 ]; then L=$R/kbbq-py_amd/kbbq/libkbbq_hip.so; else L=$R/kbbq-py_amd/kbbq/libkbbq_hip_t$T.so; fi
echo "--- round $ROUND, $T threads: pairs 4-bit / mixed"
KBBQ_HIP_LIB=$L timeout -k 10 200 python scripts/time_kernels.py --reads 50000000 --packed 2>&1 | tail -1
KBBQ_HIP_LIB=$L timeout -k 10 300 python scripts/time_mixed.py 2>&1 | tail -1 | cut -c330-470
done
done
