#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-sampling --no-roofline"
run() { label=$1; shift
  ( env "$@" timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$label', d['ms_per_step'])" ) >> gpurun_out/c13_bench.txt 2>/dev/null
}
rm -f gpurun_out/c13_bench.txt
run "default(stream32 at L=1024)" X=1
run "long16" MDM_HIP_ATTN_BWD=long16
run "default" X=1
run "long16" MDM_HIP_ATTN_BWD=long16
cat gpurun_out/c13_bench.txt
( timeout 300 python bench.py --force-collectives --wire-bf16 --bucket-mb 64 --steps 8 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-sampling --no-roofline | grep '^{' > gpurun_out/c13_force_collectives.json ) 2> gpurun_out/c13_force.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/c13_force_collectives.json').read())
    c=d['config']['comm']
    print('forced collectives ms/step', d['ms_per_step'], 'window', c.get('backward_window'))
    tl=c['bucket_timeline_ms']
    print('issue times', [t[2] for t in tl])
    print('done times', [t[3] for t in tl])
except Exception as e:
    print('force-collectives failed', e); print(open('gpurun_out/c13_force.err').read()[-2000:])
PY
( timeout 600 python -m pytest tests/test_distributed_gpu.py tests/test_trainer_gpu.py -x -q ) > gpurun_out/c13_dist_tests.txt 2>&1; tail -4 gpurun_out/c13_dist_tests.txt
