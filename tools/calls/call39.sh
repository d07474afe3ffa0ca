#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -k "attention" ) 2>&1 | tail -2
( timeout 400 python bench.py --steps 20 --warmup 5 ) > gpurun_out/c39_bench_default.json 2> gpurun_out/c39_bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/c39_bench_default.json') if l.startswith('{')][-1])
print('ms_per_step', d['ms_per_step'], 'value', d['value'], d['binary']['git_describe'])
r=d.get('roofline',{})
print('roofline', {k:r.get(k) for k in ('kernel','achieved','frac','avg_launch_ms','traffic','traffic_source','gemm_weighted')})
print('nested256', {k:v for k,v in d.get('nested256',{}).items() if k in ('ms_per_step','value')})
s=d.get('sampling',{}); print('sampling', s.get('ms_per_denoise_step'), s.get('fp32_bf16x3',{}).get('ms_per_denoise_step'))
s=d.get('nested1024_sampling',{}); print('nested1024', s.get('ms_per_denoise_step'), s.get('fp32_bf16x3',{}).get('ms_per_denoise_step'))
print([k for k in r.get('all_gemm_kernels',{}) if 'attn_fwd' in k])
PY
