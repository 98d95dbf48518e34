#!/bin/bash
# host-thread scaling of the file path (KBBQ_HOST_THREADS) on the GPU box
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc
python tests/tools/e2e_cli.py --reads 8000000 --reps 1 --keep 2>&1 | tail -3
for t in 8 16 32 64 128; do
  echo "== KBBQ_HOST_THREADS=$t"
  KBBQ_HOST_THREADS=$t KBBQ_TIMING=1 PYTHONPATH=kbbq-py_amd python -m kbbq.main recalibrate -f /tmp/e2e_a.fq /tmp/e2e_b.fq 2>&1 > /tmp/e2e_out.fq | grep stages
done
rm -f /tmp/e2e_a.fq /tmp/e2e_b.fq /tmp/e2e_out.fq
