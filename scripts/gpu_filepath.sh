# usage (GPU box): bash scripts/gpu_filepath.sh -- the file-path tests, then the bench's in-process file-path entry twice
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_pairs.py -x -q -m gpu > gpurun_out/gputests_fp.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/gputests_fp.log
timeout -k 10 600 python - <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
import bench, torch
sys.path.insert(0, os.path.join(os.getcwd(), 'kbbq-py_amd'))
from kbbq import _device as dev
dev.warm_up()
for i in range(3):
    r = bench.extra_file_path(torch, dev, n=8_000_000)
    print(round(r['value'] / 1e9, 3), r['wall_s'], r['stages_s'], flush=True)
PY
