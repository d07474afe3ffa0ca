#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_distributed_gpu.py tests/test_trainer_gpu.py -x -q ) > gpurun_out/c14_dist_tests.txt 2>&1; tail -4 gpurun_out/c14_dist_tests.txt
B="--steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-sampling --no-roofline"
( timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('plain', d['ms_per_step'])" ) 2>/dev/null
( timeout 300 python bench.py --force-collectives --wire-bf16 --bucket-mb 64 $B | grep '^{' > gpurun_out/c14_force_collectives.json ) 2> gpurun_out/c14_force.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/c14_force_collectives.json').read())
    c=d['config']['comm']
    print('forced collectives (bf16 wire, 64 MiB) ms/step', d['ms_per_step'], 'window', c.get('backward_window'))
    tl=c['bucket_timeline_ms']
    print('issue times', [t[2] for t in tl])
except Exception as e:
    print('force-collectives failed', e); print(open('gpurun_out/c14_force.err').read()[-2000:])
PY
( timeout 300 python bench.py --force-collectives --wire-fp32 --bucket-mb 256 $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('forced fp32 wire 256 MiB', d['ms_per_step'], d['config']['comm']['backward_window'])" ) 2>/dev/null
( timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('plain again', d['ms_per_step'])" ) 2>/dev/null
