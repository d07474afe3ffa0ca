#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
for lib in new old; do
  if [ $lib = old ]; then export MDM_HIP_LIB=/root/repo/ml-mdm_amd/mdm_hip/lib_oldbias.so; fi
  for shp in 0 1 3; do
    rm -rf /tmp/prof
    ( cd /tmp && KB_ONLY="$shp" KB_DBIAS=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o t -- python /root/repo/tools/kbench.py wgrad ) > /root/repo/gpurun_out/c26_prof.log 2>&1; tail -3 /root/repo/gpurun_out/c26_prof.log
    python - "$lib" "$shp" <<'PY'
import csv, glob, sys, collections
f = glob.glob('/tmp/prof/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
seq = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    if 'wgrad' in n:
        seq[n.split('(')[0][:60]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print(sys.argv[1], sys.argv[2])
for n, d in seq.items():
    h = len(d) // 2
    a, b = sorted(d[:h]), sorted(d[h:])
    print('   %-60s n=%d  no-bias median %.1f us   with-bias median %.1f us' % (n, len(d), a[len(a)//2], b[len(b)//2]))
PY
  done
done
unset MDM_HIP_LIB
( KB_DBIAS=1 timeout 200 python tools/kbench.py wgrad ) 2>&1 | tee gpurun_out/c26_wgrad_dbias.txt
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline --no-sampling"
for i in 1 2; do
( timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step new', d['ms_per_step'])" ) 2>/dev/null
( MDM_HIP_LIB=ml-mdm_amd/mdm_hip/lib_oldbias.so timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step old bias sums', d['ms_per_step'])" ) 2>/dev/null
done
