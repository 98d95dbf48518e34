# usage (GPU box): bash scripts/gpu_r3q.sh -- the host-buffer entry points slab by slab: their tests, then their PCIe-inclusive rate
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "host_buffer" > gpurun_out/gputests_r3q.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/gputests_r3q.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python scripts/time_host_entry.py 10000000 2>&1 | tee gpurun_out/host_entry.txt
for mb in 32 192; do echo "KBBQ_STAGE_MB=$mb"; KBBQ_STAGE_MB=$mb timeout -k 10 600 python scripts/time_host_entry.py 10000000 2>&1 | grep slabs | tee -a gpurun_out/host_entry.txt; done
