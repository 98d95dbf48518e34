# usage (GPU box): bash scripts/gpu_packed_long.sh -- 4-bit planes for reads of up to 320 bases in the file path: tests, then config 5 with / without
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_pairs.py tests/test_gpu_parity.py tests/test_gpu_layouts.py -x -q -m gpu > gpurun_out/gputests_pl.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/gputests_pl.log
for ROUND in 1 2; do
timeout -k 10 300 python scripts/time_mixed.py 2>&1 | tail -1 | cut -c170-900
KBBQ_PACKED_READS=160 timeout -k 10 300 python scripts/time_mixed.py 2>&1 | tail -1 | cut -c170-900
done
timeout -k 10 300 python tests/tools/fuzz_gpu_cli.py --seconds 100 > gpurun_out/fuzz_cli_pl.log 2>&1; echo "fuzz cli rc=$?"; tail -1 gpurun_out/fuzz_cli_pl.log
