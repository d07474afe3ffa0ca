#!/bin/bash
# round 6, call 12: per-shape tables of the current build (serial mode: every kernel uncontended) + the default bench line
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python tools/shape_profile.py unet64 --serial ) > gpurun_out/c12_shapes_unet64_serial.txt 2>&1
( timeout 300 python tools/shape_profile.py unet64 ) > gpurun_out/c12_shapes_unet64.txt 2>&1
( timeout 300 python tools/shape_profile.py nested256 ) > gpurun_out/c12_shapes_nested256.txt 2>&1
( timeout 500 python bench.py --steps 20 --warmup 5 ) > gpurun_out/c12_bench_default.json 2> gpurun_out/c12_bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/c12_bench_default.json') if l.startswith('{')][-1])
print('ms_per_step', d['ms_per_step'], 'value', d['value'])
r=d.get('roofline',{})
print('roofline', {k:r.get(k) for k in ('kernel','achieved','frac','avg_launch_ms','gemm_weighted')})
print('nested256', {k:v for k,v in d.get('nested256',{}).items() if k in ('ms_per_step','value')})
s=d.get('sampling',{}); print('sampling', s.get('ms_per_denoise_step'), s.get('fp32_bf16x3',{}).get('ms_per_denoise_step'))
s=d.get('nested1024_sampling',{}); print('nested1024', s.get('ms_per_denoise_step'), s.get('fp32_bf16x3',{}).get('ms_per_denoise_step'))
PY
head -30 gpurun_out/c12_shapes_unet64_serial.txt; grep -A12 "HBM-class" gpurun_out/c12_shapes_unet64_serial.txt
