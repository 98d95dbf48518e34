# usage (GPU box): bash scripts/gpu_k2_char_probe.sh -- the short-lived K2 on CHARACTER planes (one read per row, 150 bp): steps per wave / workgroup size.
# Second builds: make -C kbbq-py_amd/csrc OUT=../kbbq/libkbbq_hip_s3.so EXTRA=-DK2T_STEPS=3 (s2: 2 steps; t512: -DK2T_THREADS=512); alternating processes.
mkdir -p gpurun_out
D=$GRAFT_REPO_ROOT/kbbq-py_amd/kbbq
for round in 1 2 3; do
  for v in shipped s3 s2 t512; do
    L=$D/libkbbq_hip_$v.so; [ $v = shipped ] && L=$D/libkbbq_hip.so
    [ -f $L ] || continue
    echo "round $round $v : $(KBBQ_HIP_LIB=$L timeout -k 10 200 python scripts/time_kernels.py --reads 50000000 2>&1 | tail -1)"
  done
done | tee gpurun_out/k2_char_probe.log
