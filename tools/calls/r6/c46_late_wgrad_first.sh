#!/bin/bash
# the late-gradient layers' weight gradients handed to the side stream before their input-gradient launches: A/B, tests
cd /root/repo
export TMPDIR=/tmp
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested1024 --no-roofline --no-sampling"
for i in 1 2 3; do
for m in product old; do
( if [ $m = old ]; then export MDM_HIP_LATE_WGRAD_FIRST=0; fi; timeout 300 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step variant=$m', d['ms_per_step'], 'nested256', d['nested256']['ms_per_step'])" ) 2>&1 | tail -1
done
done
timeout 1500 python -m pytest tests/test_trainer_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "text or shared or linears or train or fused or kv" 2>&1 | tail -3
