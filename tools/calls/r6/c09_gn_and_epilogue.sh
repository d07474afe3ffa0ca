#!/bin/bash
# round 6, call 9: (a) GroupNorm kernels whose operand loads are batched / pipelined ahead of the stores vs the round-5 norm.hip;
# (b) epilogue variants V0 .. V3 of the GEMM vs the round-5 epilogue
cd /root/repo
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "group_norm or gn" ) 2>&1 | tail -3
for m in new normold; do
echo "--- norm=$m (cold buffers, + residual gradient)"
( if [ $m = normold ]; then export MDM_HIP_LIB=/root/repo/ab_libs/lib_normold.so; fi; KB_GN_COLD=8 KB_GN_RES=1 timeout 300 python tools/kbench.py gn 2>&1 | grep "^gn" )
done
for m in v0 v1 v2 v3 r5; do
echo "--- variant=$m"
( export MDM_HIP_LIB=/root/repo/ab_libs/lib_$m.so; if [ $m = r5 ]; then export MDM_HIP_LIB=/root/repo/ab_libs/lib_r5gelu.so; fi; timeout 250 python tools/kbench.py rotate 2>&1 | grep "fresh" | grep "768->3072\|768->2304\|512->1536" | sed 's/| same.*//' )
done
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline --no-sampling"
for i in 1 2 3; do
for m in v0 v1 v2 v3 r5 normold product; do
( if [ $m != product ]; then export MDM_HIP_LIB=/root/repo/ab_libs/lib_$m.so; fi; if [ $m = r5 ]; then export MDM_HIP_LIB=/root/repo/ab_libs/lib_r5gelu.so; fi; timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step variant=$m', d['ms_per_step'])" ) 2>&1 | tail -1
done
done
