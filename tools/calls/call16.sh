#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
( timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r5/bench_default.json ) 2> gpurun_out/r5/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5/bench_default.json') if l.startswith('{')][-1])
print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'binary', d.get('binary'))
r=d['roofline']; print('roofline', {k:r[k] for k in ('kernel','achieved','frac','avg_launch_ms','traffic')}, r['gemm_weighted'])
print('nested256', {k:v for k,v in d.get('nested256',{}).items() if k in ('ms_per_step','value','step_mfma_roofline_frac')})
s=d.get('sampling'); print('sampling', {k:s[k] for k in s if k.startswith('ms_')}, s.get('fp32_bf16x3'))
print('nested1024', {k:v for k,v in d.get('nested1024_sampling',{}).items() if 'ms_per' in k or k=='fp32_bf16x3'})
print('reference_loop', d.get('reference_loop'))
print('cpu_baseline', d.get('cpu_baseline'))
PY
( timeout 200 python tools/shape_profile.py unet64 > gpurun_out/r5/shapes_unet64.txt ) 2>/dev/null
( timeout 200 python tools/shape_profile.py unet64 --serial > gpurun_out/r5/shapes_unet64_serial.txt ) 2>/dev/null
( timeout 200 python tools/shape_profile.py nested256 > gpurun_out/r5/shapes_nested256.txt ) 2>/dev/null
grep -i "attn\|GEMM-class" gpurun_out/r5/shapes_unet64.txt gpurun_out/r5/shapes_unet64_serial.txt gpurun_out/r5/shapes_nested256.txt
cd /tmp
( timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /root/repo/gpurun_out/r5/pmc_f -o pmc -- python /root/repo/tools/attn_debug.py pmc ) > /dev/null 2>&1
( timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /root/repo/gpurun_out/r5/pmc_w -o pmc -- python /root/repo/tools/attn_debug.py pmc ) > /dev/null 2>&1
cd /root/repo
python tools/pmc_traffic.py gpurun_out/r5/pmc_f gpurun_out/r5/pmc_w gpurun_out/r5/pmc_attention_traffic.json > /dev/null 2>&1
rm -rf gpurun_out/r5/pmc_f gpurun_out/r5/pmc_w
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5/pmc_attention_traffic.json'))
ks=d.get('kernels', d)
for k,v in ks.items():
    if 'attn' in k: print(k[:70], v)
PY
