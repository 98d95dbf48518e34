#!/bin/bash
# A/B builds of K2 at different occupancies / prefetch depths (one job, one device)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/k2occ
run() { # name, EXTRA
  make -s -C kbbq-py_amd/csrc OUT=$GRAFT_REPO_ROOT/gpurun_out/k2occ/lib_$1.so EXTRA="$2" > gpurun_out/k2occ/build_$1.log 2>&1 || { echo "build $1 failed"; tail -3 gpurun_out/k2occ/build_$1.log; return; }
  printf '%-28s ' "$1 [$2]"; KBBQ_HIP_LIB=$GRAFT_REPO_ROOT/gpurun_out/k2occ/lib_$1.so timeout -k 10 120 python scripts/time_kernels.py --pairs 2>&1 | grep ABLATE | sed 's/.*len=150: //'
  rm -f gpurun_out/k2occ/lib_$1.so
}
run base ""
run w2 "-DK2V3_WAVES=2"
run w2n3 "-DK2V3_WAVES=2 -DK2V3_NBUF=3"
run w2n4 "-DK2V3_WAVES=2 -DK2V3_NBUF=4"
run t256w2 "-DK2V3_THREADS=256 -DK2V3_WAVES=2"
run t256w2n3 "-DK2V3_THREADS=256 -DK2V3_WAVES=2 -DK2V3_NBUF=3"
run w3 "-DK2V3_WAVES=3 -DK2V3_THREADS=768"
run base2 ""
