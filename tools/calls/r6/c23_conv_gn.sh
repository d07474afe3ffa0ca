#!/bin/bash
# round 6, call 23: the opt-in convolution + GroupNorm launch (proj_out -> ffn[0], round 3: step-neutral) re-tried on the fixed epilogue
cd /root/repo
export TMPDIR=/tmp
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline --no-sampling"
for i in 1 2 3; do
for m in 0 1; do
( MDM_HIP_CONV_GN=$m timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step conv_gn=$m', d['ms_per_step'])" ) 2>&1 | tail -1
done
done
