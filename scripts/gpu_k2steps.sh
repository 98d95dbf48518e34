# usage (GPU box): bash scripts/gpu_k2steps.sh -- the short-lived K2's tile geometry once more under the 8-front order: KiB-steps per wave 2 / 3 / 4 (shipped) / 6
# and 512-thread workgroups.  Second builds beforehand:  for n in 2 3 6: make -C kbbq-py_amd/csrc OUT=../kbbq/libkbbq_hip_s$n.so EXTRA="-DK2T_STEPS=$n";
#                                                        make -C kbbq-py_amd/csrc OUT=../kbbq/libkbbq_hip_t512.so EXTRA="-DK2T_THREADS=512"
R=$GRAFT_REPO_ROOT
cd $R
for ROUND in 1 2 3; do
for V in base s2 s3 s6 t512; do
if [ $V = base ]; then L=$R/kbbq-py_amd/kbbq/libkbbq_hip.so; else L=$R/kbbq-py_amd/kbbq/libkbbq_hip_$V.so; fi
echo "round $ROUND $V: $(KBBQ_HIP_LIB=$L timeout -k 10 200 python scripts/time_kernels.py --reads 50000000 --packed 2>&1 | tail -1)"
done
done 2>&1 | tee gpurun_out/k2steps.txt
