# Kernel timeline of a few bench steps (start / end of every kernel): where the chip idles between launches.
# usage: [TRACE_ENV="VAR=val ..."] gpu_trace.sh TAG  -> gpurun_out/TAG_trace.csv.gz (kernel name, queue, start, end, ...)
mkdir -p gpurun_out
T=${1:-trace}
export TMPDIR=/tmp
D=/tmp/trace_$T
rm -rf $D
env $TRACE_ENV rocprofv3 --kernel-trace --output-format csv -d $D -- python bench.py --steps 4 --warmup 2 --no_cpu_baseline --no_other_configs > gpurun_out/${T}_trace_bench.log 2>&1
f=$(find $D -name "*kernel_trace.csv" | head -1)
echo "trace file: $f $(wc -l < $f) rows"
python - "$f" gpurun_out/${T}_trace.csv.gz <<'PY'
import csv, gzip, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
with gzip.open(sys.argv[2], 'wt') as g:
  w = csv.writer(g)
  w.writerow(['name', 'queue', 'stream', 'start', 'end', 'lds', 'vgpr', 'grid', 'wg', 'dispatch'])
  for r in rows:
    w.writerow([r['Kernel_Name'][:120], r.get('Queue_Id', ''), r.get('Stream_Id', ''), r['Start_Timestamp'], r['End_Timestamp'], r['LDS_Block_Size'], r['VGPR_Count'], r['Grid_Size_X'], r['Workgroup_Size_X'], r['Dispatch_Id']])
PY
tail -1 gpurun_out/${T}_trace_bench.log | cut -c1-200
ls -la gpurun_out/${T}_trace.csv.gz
