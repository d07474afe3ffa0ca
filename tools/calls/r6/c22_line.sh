#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r6
export TMPDIR=/tmp
( timeout 500 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r6/bench_default2.json 2> gpurun_out/r6/bench_default2.err
python - <<'PY'
import json
d=json.loads([l for l in open('/root/repo/gpurun_out/r6/bench_default2.json') if l.startswith('{')][-1])
print('ms_per_step', d['ms_per_step'], 'value', d['value'], d['binary']['sha256'][:16], d['binary'].get('git_describe'))
r=d.get('roofline',{})
print('roofline', {k:r.get(k) for k in ('kernel','achieved','frac','avg_launch_ms','traffic','traffic_source','gemm_weighted')})
print('hbm', json.dumps(r.get('hbm_kernels'))[:1200])
print('nested256', {k:v for k,v in d.get('nested256',{}).items() if k in ('ms_per_step','value')})
s=d.get('sampling',{}); print('sampling', s.get('ms_per_denoise_step'), s.get('fp32_bf16x3',{}).get('ms_per_denoise_step'))
s=d.get('nested1024_sampling',{}); print('nested1024', s.get('ms_per_denoise_step'), s.get('fp32_bf16x3',{}).get('ms_per_denoise_step'), s.get('hbm'))
print('reference_loop', d.get('reference_loop'))
PY
