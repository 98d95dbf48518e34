# usage (GPU box): bash scripts/gpu_write_bench.sh -- how fast can 2.5 GB of rendered text reach a fresh file: write(2) of 80 MB slabs (mode 0),
# ftruncate + mmap + memcpy on N threads (mode 1), posix_fallocate + the same (mode 2); in the directories the command line's output goes to.
mkdir -p gpurun_out
g++ -O2 -pthread -D_GNU_SOURCE -o /tmp/write_bench scripts/write_bench.cpp || exit 1
for d in /tmp "$GRAFT_REPO_ROOT/gpurun_out" /dev/shm; do
  echo "== $d ($(df -T $d | tail -1 | awk '{print $2}'))"
  for rep in 1 2; do
    /tmp/write_bench $d/wb.bin 0 1 2500
    for nt in 4 8 16; do /tmp/write_bench $d/wb.bin 1 $nt 2500; done
    /tmp/write_bench $d/wb.bin 2 16 2500
    /tmp/write_bench $d/wb.bin 3 1 2500
    /tmp/write_bench $d/wb.bin 4 1 2500
  done
done 2>&1 | tee gpurun_out/write_bench.log
