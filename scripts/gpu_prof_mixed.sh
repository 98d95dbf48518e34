# usage (GPU box): bash scripts/gpu_prof_mixed.sh TAG -- kernel trace of the mixed-length entry: per-band K1 / K2 durations
TAG=${1:-mixed}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/stats_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $OUT -o s --output-format csv -- python $R/scripts/time_mixed.py > $R/gpurun_out/mixed_$TAG.json 2> $R/gpurun_out/mixed_$TAG.err; echo "rocprof rc=$?"
python - <<PY
import csv, glob, collections
for f in glob.glob('$OUT/**/s_kernel_trace.csv', recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if r['Kernel_Name'].startswith(('void k1v3', 'void k2v3', 'k2t_', 'void k1_', 'void k2_'))]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    for r in rows[-32:]:
        print('%-52s grid %9s wg %5s lds %6s vgpr %3s  %8.1f us' % (r['Kernel_Name'][:52], r['Grid_Size_X'], r['Workgroup_Size_X'], r['LDS_Block_Size'], r['VGPR_Count'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
PY
