#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "attention" ) 2>&1 | tail -2
( timeout 900 python -m pytest tests/test_model_gpu.py -x -q ) 2>&1 | tail -2
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline --no-sampling"
for i in 1 2; do
for m in default fwd16 fwd32; do
( if [ $m != default ]; then export MDM_HIP_ATTN_FWD=$m; fi; timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step forward=$m', d['ms_per_step'])" ) 2>/dev/null
done
done
