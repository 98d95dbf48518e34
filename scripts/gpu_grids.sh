# usage (GPU box): bash scripts/gpu_grids.sh -- launch geometry of the streaming kernels with grid-stride loops
# (workgroups per CU; 0 = one work item per thread): K4 / K5 / K6 on 16 M aligned reads, k7_lay_out on 50 M reads;
# three interleaved rounds (devices and clocks drift by several per cent between processes)
R=$GRAFT_REPO_ROOT
cd $R
for ROUND in 1 2 3; do
for G in default 32 64 128 256; do
  echo "--- round $ROUND K4/K5/K6 grids = $G"
  if [ $G = default ]; then timeout -k 10 300 python scripts/time_benchmark_path.py --reads 16000000 --ins 0.05 2>&1 | grep "flags plane\|4-bit" | cut -c1-60
  else KBBQ_K4_GRID=$G KBBQ_K5_GRID=$G KBBQ_K6_GRID=$G timeout -k 10 300 python scripts/time_benchmark_path.py --reads 16000000 --ins 0.05 2>&1 | grep "flags plane\|4-bit" | cut -c1-60; fi
done
for G in default 64 256 1024; do
  echo "--- round $ROUND k7_lay_out grid = $G"
  if [ $G = default ]; then timeout -k 10 300 python scripts/time_layout.py 2>&1 | grep "packed=True" | tail -1 | cut -c1-50
  else KBBQ_K7_GRID=$G timeout -k 10 300 python scripts/time_layout.py 2>&1 | grep "packed=True" | tail -1 | cut -c1-50; fi
done
done
