#!/bin/bash
# fixed per-process cost of the CLI on the GPU box
export PYTHONPATH=kbbq-py_amd
TIMEFORMAT='%R s'
t() { printf '%-75s ' "$*"; time "$@" > /dev/null 2>&1; }
t python -c "pass"
t python -c "import numpy"
t python -c "import torch"
t python -c "import torch"
t python -c "import scipy.special"
t python -c "import kbbq.main"
t python -c "import torch; torch.cuda.init(); torch.zeros(1, device='cuda')"
t python -c "from kbbq import _device as d; d.context()"
t python -c "import torch, time; from kbbq import _device as d; d.context(); import os; os._exit(0)"
python tests/tools/e2e_cli.py --reads 8000000 --reps 1 --keep 2>&1 | grep -v amdgpu.ids
KBBQ_TIMING=1 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import time, sys, os
t0 = time.perf_counter()
import torch
t1 = time.perf_counter()
from kbbq import main, recalibrate, _device as dev
t2 = time.perf_counter()
dev.context(); t3 = time.perf_counter()
fd = os.open('/tmp/e2e_out.fq', os.O_WRONLY | os.O_CREAT | os.O_TRUNC); saved = os.dup(1); os.dup2(fd, 1)
recalibrate.recalibrate_fastq(['/tmp/e2e_a.fq', '/tmp/e2e_b.fq'])
sys.stdout.flush()
t4 = time.perf_counter()
import gc; gc.collect(); torch.cuda.synchronize(); t5 = time.perf_counter()
from kbbq import _trace; _trace.report()
os.lseek(1, 0, os.SEEK_SET)
t6 = time.perf_counter()
recalibrate.recalibrate_fastq(['/tmp/e2e_a.fq', '/tmp/e2e_b.fq'])
sys.stdout.flush()
t7 = time.perf_counter()
print('import torch %.2f  import kbbq %.2f  context %.2f  recalibrate_fastq %.2f  gc %.2f  second call %.2f'
      % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t7 - t6), file=sys.stderr)
PY
rm -f /tmp/e2e_a.fq /tmp/e2e_b.fq /tmp/e2e_out.fq
