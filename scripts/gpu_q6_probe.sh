# usage (GPU box): bash scripts/gpu_q6_probe.sh -- TIMING ONLY: what would K1 / K2 gain from qualities packed 16 to 12 bytes (6 bits each)?
# A second build (make -C kbbq-py_amd/csrc OUT=../kbbq/libkbbq_hip_q6probe.so EXTRA=-DKBBQ_Q6_PROBE) loads 12 instead of 16 quality bytes per
# chunk and does the unpacking work on whatever lies there (wrong results); alternating processes on one device.
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/kbbq-py_amd/kbbq/libkbbq_hip_q6probe.so
for round in 1 2 3; do
  echo "round $round shipped : $(timeout -k 10 200 python scripts/time_kernels.py --reads 50000000 --packed 2>&1 | tail -1)"
  echo "round $round 6-bit   : $(KBBQ_HIP_LIB=$L timeout -k 10 200 python scripts/time_kernels.py --reads 50000000 --packed 2>&1 | tail -1)"
done | tee gpurun_out/q6_probe.log
