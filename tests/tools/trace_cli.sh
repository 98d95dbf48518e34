# usage (GPU box): bash tests/tools/trace_cli.sh TAG (test infrastructure: its input files come from the oracle's generator) -- rocprofv3 kernel trace of `kbbq recalibrate -f A B --infer-rg` on an 8-read-group input:
# which kernels does the product path launch?
TAG=${1:-cli}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/trace_$TAG
mkdir -p $OUT
cd $R
python - <<PY
import sys
sys.path.insert(0, 'oracle'); sys.path.insert(0, 'tests')
import oracle as O
n, nrg = 200000, 8
seq, cseq, qual, meta = O.synth(0, n, n, 3, 150, 150, nrg)
names = O.synth_names(0, n, nrg, with_rg=True)
O.write_fastq('/tmp/trace_a.fq', names, seq, qual, meta)
O.write_fastq('/tmp/trace_b.fq', names, cseq, qual, meta)
PY
cd /tmp && export TMPDIR=/tmp
cat > /tmp/run_cli.py <<PY
import sys
sys.path.insert(0, '$R/kbbq-py_amd')
from kbbq import main
sys.argv = ['kbbq', 'recalibrate', '-f', '/tmp/trace_a.fq', '/tmp/trace_b.fq', '--infer-rg', '-o', '/tmp/trace_out.fq']
main.main()
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT -o s --output-format csv -- python /tmp/run_cli.py > $OUT/run.log 2>&1; echo "rocprof rc=$?"
python - <<PY
import csv, glob
for f in glob.glob('$OUT/**/s_kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print('%-90s calls %4s total %9.1f us' % (r['Name'][:90], r['Calls'], float(r['TotalDurationNs']) / 1e3))
PY
