#!/bin/bash
# one CLI run with the stage timeline (KBBQ_TIMING=2)
python tests/tools/e2e_cli.py --reads ${1:-8000000} --reps 1 --keep 2>&1 | tail -3
for rep in 1 2; do
s=$(date +%s.%N)
KBBQ_TIMING=2 PYTHONPATH=kbbq-py_amd python -m kbbq.main recalibrate -f /tmp/e2e_a.fq /tmp/e2e_b.fq 2>/tmp/tl.txt > /tmp/e2e_out.fq
e=$(date +%s.%N)
cat /tmp/tl.txt | grep -v amdgpu.ids
python -c "print('wall %.3f s' % ($e - $s))"
done
rm -f /tmp/e2e_a.fq /tmp/e2e_b.fq /tmp/e2e_out.fq
