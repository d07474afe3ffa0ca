#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( KB_DBIAS=1 timeout 200 python tools/kbench.py wgrad ) 2>&1 | tee gpurun_out/c22_wgrad_dbias.txt
( timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/c22_gpu_tests.txt 2>&1
tail -5 gpurun_out/c22_gpu_tests.txt
( timeout 300 python bench.py --steps 20 --warmup 5 ) > gpurun_out/c22_bench_default.json 2> gpurun_out/c22_bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/c22_bench_default.json') if l.startswith('{')][-1])
print('ms_per_step', d['ms_per_step'], 'value', d['value'])
print('roofline', d.get('roofline'))
print('nested256', {k:v for k,v in d.get('nested256',{}).items() if k in ('ms_per_step','value')})
print('sampling', d.get('sampling'))
print('nested1024', d.get('nested1024_sampling'))
PY
