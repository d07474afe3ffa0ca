#!/bin/bash
# final validation of the round: full GPU suite, smoke, default bench line, the step under rocprofv3, shape tables
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/c32_gpu_tests.txt 2>&1
tail -4 gpurun_out/c32_gpu_tests.txt
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -3
( timeout 400 python bench.py --steps 20 --warmup 5 ) > gpurun_out/c32_bench_default.json 2> gpurun_out/c32_bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/c32_bench_default.json') if l.startswith('{')][-1])
print('ms_per_step', d['ms_per_step'], 'value', d['value'])
r=d.get('roofline',{})
print('roofline', {k:r.get(k) for k in ('kernel','achieved','frac','avg_launch_ms','gemm_weighted')})
print('nested256', {k:v for k,v in d.get('nested256',{}).items() if k in ('ms_per_step','value')})
s=d.get('sampling',{}); print('sampling', s.get('ms_per_denoise_step'), s.get('fp32_bf16x3',{}).get('ms_per_denoise_step'))
s=d.get('nested1024_sampling',{}); print('nested1024', s.get('ms_per_denoise_step'), s.get('fp32_bf16x3',{}).get('ms_per_denoise_step'))
PY
cd /tmp
( timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/c32_prof -o bench -- python /root/repo/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-nested --no-reference-loop --no-nested1024 --no-sampling > /root/repo/gpurun_out/c32_bench_profiled.json ) 2> /root/repo/gpurun_out/c32_prof.err
cd /root/repo
DB=$(find gpurun_out/c32_prof -name "*.db" | head -1)
python tools/kstats_db.py $DB 60 --train-steps > gpurun_out/c32_kernel_stats.txt 2>&1
python tools/fwd_gaps.py $DB > gpurun_out/c32_stream_windows.txt 2>&1
rm -rf gpurun_out/c32_prof
head -12 gpurun_out/c32_kernel_stats.txt | cut -c1-150; tail -4 gpurun_out/c32_stream_windows.txt
( timeout 300 python tools/shape_profile.py unet64 ) > gpurun_out/c32_shapes_unet64.txt 2>&1
( timeout 300 python tools/shape_profile.py unet64 --serial ) > gpurun_out/c32_shapes_unet64_serial.txt 2>&1
head -3 gpurun_out/c32_shapes_unet64.txt; head -3 gpurun_out/c32_shapes_unet64_serial.txt
