export MADE_DEBUG_VARIANTS=1          # (measurement knobs are honoured only under this switch)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lintiles; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/wa; rocprofv3 --kernel-trace --output-format csv -d /tmp/wa -- python3 $R/tools/linear_tiles_bench.py > /dev/null 2>&1
echo "default dispatch"; python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/wa/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'linear' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
for i in range(0,len(d),8): print(' '.join(f'{x:6.1f}' for x in d[i:i+8]))
PY
export MADE_LINEAR_TILE=64
rm -rf /tmp/wb; rocprofv3 --kernel-trace --output-format csv -d /tmp/wb -- python3 $R/tools/linear_tiles_bench.py > /dev/null 2>&1
echo "tiled (MADE_LINEAR_TILE=64)"; python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/wb/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'linear' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
for i in range(0,len(d),8): print(' '.join(f'{x:6.1f}' for x in d[i:i+8]))
PY
export MADE_LINEAR_TILE=128
rm -rf /tmp/wc; rocprofv3 --kernel-trace --output-format csv -d /tmp/wc -- python3 $R/tools/linear_tiles_bench.py > /dev/null 2>&1
echo "tiled (MADE_LINEAR_TILE=128)"; python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/wc/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'linear' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
for i in range(0,len(d),8): print(' '.join(f'{x:6.1f}' for x in d[i:i+8]))
PY
