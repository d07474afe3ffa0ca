#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "norm" ) 2>&1 | tail -3
echo "--- gn chunked (cold)"; ( KB_GN_COLD=8 KB_GN_RES=1 timeout 300 python tools/kbench.py gn ) 2>&1 | grep "^gn 64\|^gn 32" | tee gpurun_out/c31_gn_chunk_cold.txt
echo "--- gn not chunked (cold)"; ( MDM_HIP_GN_CHUNK_MB=-1 KB_GN_COLD=8 KB_GN_RES=1 timeout 300 python tools/kbench.py gn ) 2>&1 | grep "^gn 64\|^gn 32" | tee gpurun_out/c31_gn_nochunk_cold.txt
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline --no-sampling"
for i in 1 2; do
( timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step chunked gn bwd', d['ms_per_step'])" ) 2>/dev/null
( MDM_HIP_GN_CHUNK_MB=-1 timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step not chunked', d['ms_per_step'])" ) 2>/dev/null
done
