#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-sampling --no-roofline --no-nested --no-nested1024 --no-reference-loop"
cd /tmp
( timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /root/repo/gpurun_out/pmc5_f -o pmc -- python /root/repo/bench.py $ARGS ) > /root/repo/gpurun_out/c37_f.log 2>&1
( timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /root/repo/gpurun_out/pmc5_w -o pmc -- python /root/repo/bench.py $ARGS ) > /root/repo/gpurun_out/c37_w.log 2>&1
cd /root/repo
python tools/pmc_traffic.py gpurun_out/pmc5_f gpurun_out/pmc5_w gpurun_out/c37_pmc_hbm_traffic.json 2>&1 | tail -5
rm -rf gpurun_out/pmc5_f gpurun_out/pmc5_w
python - <<'PY'
import json
d=json.load(open('gpurun_out/c37_pmc_hbm_traffic.json'))
ks=d.get('kernels',d)
tot=0
rows=[]
for k,v in ks.items():
    if isinstance(v,dict) and 'hbm_bytes_per_launch' in v:
        rows.append((v['hbm_bytes_per_launch']*v['launches'],k,v))
rows.sort(reverse=True)
for t,k,v in rows[:14]: print('%8.2f GB total  %5d launches  %8.1f MB/launch  %s' % (t/1e9, v['launches'], v['hbm_bytes_per_launch']/1e6, k[:90]))
print({k:v for k,v in d.items() if k!='kernels'})
PY
