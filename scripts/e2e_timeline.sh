#!/bin/bash
# the command line as a user runs it: wall time of the whole command (stage sums on stderr), then once more with the stage timeline
python tests/tools/e2e_cli.py --reads ${1:-8000000} --reps 1 --keep 2>&1 | tail -3
for rep in 1 2 3; do
rm -f /tmp/e2e_out.fq      # a fresh output file: truncating 2.5 GB of page cache is the kernel's 0.4 s, not the command's
s=$(date +%s.%N)
KBBQ_TIMING=1 PYTHONPATH=kbbq-py_amd python -m kbbq.main recalibrate -f /tmp/e2e_a.fq /tmp/e2e_b.fq 2>/tmp/tl.txt > /tmp/e2e_out.fq
e=$(date +%s.%N)
grep -v amdgpu.ids /tmp/tl.txt
python -c "print('wall %.3f s (whole command, interpreter start to exit)' % ($e - $s))"
done
KBBQ_TIMING=2 PYTHONPATH=kbbq-py_amd python -X importtime -m kbbq.main recalibrate -f /tmp/e2e_a.fq /tmp/e2e_b.fq 2>/tmp/tl.txt > /tmp/e2e_out.fq
grep -v "amdgpu.ids\|import time" /tmp/tl.txt | grep -v " write$\| format$\| D2H$"
echo "imports costing more than 20 ms (cumulative us | module):"
grep "import time" /tmp/tl.txt | awk -F'|' '$2+0 > 20000 {print $2 "|" $3}' | sort -n | tail -15
grep -c "| *torch" /tmp/tl.txt
rm -f /tmp/e2e_a.fq /tmp/e2e_b.fq /tmp/e2e_out.fq
