#!/bin/bash
# round 6, call 32: the final binary -- whole GPU suite, smoke, PMC traffic first (the bench line quotes it), default line,
# rocprofv3 kernel stats / stream windows, shape tables, forced collectives, clocks and power under the step
cd /root/repo
mkdir -p gpurun_out/r6
export TMPDIR=/tmp
O=/root/repo/gpurun_out/r6
sha256sum ml-mdm_amd/mdm_hip/libmdm_hip.so | cut -c1-64 > $O/binary_sha256.txt
( timeout 2400 python -m pytest tests -m gpu -q ) > $O/gpu_tests.txt 2>&1
tail -3 $O/gpu_tests.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -2
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-sampling --no-roofline --no-nested --no-nested1024 --no-reference-loop"
cd /tmp
( timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o pmc -- python /root/repo/bench.py $ARGS ) > $O/pmc_f.log 2>&1
( timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o pmc -- python /root/repo/bench.py $ARGS ) > $O/pmc_w.log 2>&1
cd /root/repo
python tools/pmc_traffic.py $O/pmc_f $O/pmc_w $O/pmc_hbm_traffic.json 2>&1 | tail -3
rm -rf $O/pmc_f $O/pmc_w
cd /tmp
for n in 2 6; do
( timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/s_f$n -o pmc -- python /root/repo/tools/sample_pmc.py run $n ) > $O/s_f$n.log 2>&1
( timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/s_w$n -o pmc -- python /root/repo/tools/sample_pmc.py run $n ) > $O/s_w$n.log 2>&1
done
cd /root/repo
python tools/sample_pmc.py sum $O/s_f2 $O/s_w2 $O/s_f6 $O/s_w6 2 6 $O/pmc_nested1024_sampling.json 2>&1 | tail -2
rm -rf $O/s_f2 $O/s_w2 $O/s_f6 $O/s_w6
# the line quotes the traffic files of THIS binary
[ -s $O/pmc_hbm_traffic.json ] && cp $O/pmc_hbm_traffic.json profiles/r06_pmc_hbm_traffic.json
[ -s $O/pmc_nested1024_sampling.json ] && cp $O/pmc_nested1024_sampling.json profiles/r06_pmc_nested1024_sampling.json
( timeout 600 python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
cd /tmp
( timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python /root/repo/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-nested --no-reference-loop --no-nested1024 --no-sampling > $O/bench_profiled.json ) 2> $O/prof.err
cd /root/repo
DB=$(find $O/prof -name "*.db" | head -1)
python tools/kstats_db.py $DB 70 --train-steps > $O/kernel_stats.txt 2>&1
python tools/fwd_gaps.py $DB > $O/stream_windows.txt 2>&1
rm -rf $O/prof
( timeout 300 python tools/shape_profile.py unet64 ) > $O/shapes_unet64.txt 2>&1
( timeout 300 python tools/shape_profile.py unet64 --serial ) > $O/shapes_unet64_serial.txt 2>&1
( timeout 300 python tools/shape_profile.py nested256 ) > $O/shapes_nested256.txt 2>&1
( timeout 300 python bench.py --force-collectives --bucket-mb 64 --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline --no-sampling ) > $O/bench_force_collectives.json 2> /dev/null
( timeout 300 python tools/clock_watch.py unet64 12 ) > $O/clock_power.txt 2>&1
python - <<'PY'
import json
O='/root/repo/gpurun_out/r6/'
d=json.loads([l for l in open(O+'bench_default.json') if l.startswith('{')][-1])
print('ms_per_step', d['ms_per_step'], 'value', d['value'], d['binary']['sha256'][:16])
r=d.get('roofline',{})
print('roofline', {k:r.get(k) for k in ('kernel','achieved','frac','avg_launch_ms','traffic','gemm_weighted')})
print('nested256', {k:v for k,v in d.get('nested256',{}).items() if k in ('ms_per_step','value')})
s=d.get('sampling',{}); print('sampling', s.get('ms_per_denoise_step'), s.get('fp32_bf16x3',{}).get('ms_per_denoise_step'), s.get('fp32_bf16x3',{}).get('ratio_to_bf16'))
s=d.get('nested1024_sampling',{}); print('nested1024', s.get('ms_per_denoise_step'), s.get('fp32_bf16x3',{}).get('ms_per_denoise_step'), s.get('hbm'))
print('cpu_baseline', json.dumps(d.get('cpu_baseline'))[:700])
f=json.loads([l for l in open(O+'bench_force_collectives.json') if l.startswith('{')][-1])
print('forced collectives', f['ms_per_step'], f['config']['comm'].get('backward_window'))
PY
head -14 $O/kernel_stats.txt | cut -c1-150; tail -6 $O/stream_windows.txt; head -3 $O/shapes_unet64_serial.txt; head -5 $O/pmc_nested1024_sampling.json; tail -3 $O/clock_power.txt
