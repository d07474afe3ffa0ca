#!/bin/bash
# round 6, call 15: GroupNorm backward at 16x16 as 8 passes of 256-thread blocks (one round of 768 blocks) vs 4 passes of 512
cd /root/repo
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "group_norm or gn" ) 2>&1 | tail -2
for m in 1 0; do
echo "--- MDM_HIP_GN_BWD256=$m (cold buffers, + residual gradient)"
( MDM_HIP_GN_BWD256=$m KB_GN_COLD=8 KB_GN_RES=1 timeout 300 python tools/kbench.py gn 2>&1 | grep "^gn 16" )
echo "--- MDM_HIP_GN_BWD256=$m (cold, FiLM + two residual gradients)"
( MDM_HIP_GN_BWD256=$m KB_GN_COLD=8 KB_GN_RES=1 KB_GN_FULL=1 timeout 300 python tools/kbench.py gn 2>&1 | grep "^gn 16" )
done
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline --no-sampling"
for i in 1 2 3; do
for m in 1 0; do
( MDM_HIP_GN_BWD256=$m timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step gn_bwd256=$m', d['ms_per_step'])" ) 2>&1 | tail -1
done
done
