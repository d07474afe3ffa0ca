#!/bin/bash
# round 6, call 10: the settled epilogue (no loads in the store loop) + pipelined GroupNorm kernels: ops tests, GN timing, step
cd /root/repo
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_diffusion_ops_gpu.py -x -q ) 2>&1 | tail -3
for m in new normold; do
echo "--- norm=$m (cold buffers, + residual gradient)"
( if [ $m = normold ]; then export MDM_HIP_LIB=/root/repo/ab_libs/lib_normold.so; fi; KB_GN_COLD=8 KB_GN_RES=1 timeout 300 python tools/kbench.py gn 2>&1 | grep "^gn" )
done
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline --no-sampling"
for i in 1 2; do
for m in product r5; do
( if [ $m = r5 ]; then export MDM_HIP_LIB=/root/repo/ab_libs/lib_r5gelu.so; fi; timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step variant=$m', d['ms_per_step'])" ) 2>&1 | tail -1
done
done
