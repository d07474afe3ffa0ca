#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
# 1. the forced-collectives timeline after the late-parameter ordering
( timeout 300 python bench.py --force-collectives --wire-bf16 --bucket-mb 64 --steps 8 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-sampling --no-roofline | grep '^{' > gpurun_out/c12_force_collectives.json ) 2> gpurun_out/c12_force.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/c12_force_collectives.json').read())
    c=d['config']['comm']
    print('forced collectives ms/step', d['ms_per_step'], 'window', c.get('backward_window'))
    tl=c['bucket_timeline_ms']
    print('buckets', len(tl), 'MB', [t[1] for t in tl])
    print('issue times', [t[2] for t in tl])
    print('done times', [t[3] for t in tl])
except Exception as e:
    print('force-collectives failed', e); print(open('gpurun_out/c12_force.err').read()[-2000:])
PY
# 2. plain bench right after, same box (what the collectives cost in a world of one)
( timeout 250 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-sampling --no-roofline | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('plain', d['ms_per_step'])" ) 2>/dev/null
# 3. the step under rocprofv3: kernel table of the training steps + stream windows
cd /tmp
( timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/c12_prof -o bench -- python /root/repo/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-nested --no-reference-loop --no-nested1024 --no-sampling > /root/repo/gpurun_out/c12_bench_profiled.json ) 2> /root/repo/gpurun_out/c12_prof.err
cd /root/repo
DB=$(find gpurun_out/c12_prof -name "*.db" | head -1)
echo "db: $DB"
python tools/kstats_db.py $DB 60 --train-steps > gpurun_out/c12_kernel_stats.txt 2>&1
python tools/fwd_gaps.py $DB > gpurun_out/c12_stream_windows.txt 2>&1
rm -rf gpurun_out/c12_prof
head -30 gpurun_out/c12_kernel_stats.txt; tail -12 gpurun_out/c12_stream_windows.txt
