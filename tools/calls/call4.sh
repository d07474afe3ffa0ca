#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python tools/attn_debug.py time ) > gpurun_out/c4_attn_debug.txt 2>&1
( timeout 120 python tools/attn_debug.py stamps ) > gpurun_out/c4_stamps.txt 2>&1
( timeout 400 python -m pytest tests/test_ops_gpu.py -k "test_attention" -q ) > gpurun_out/c4_attn_tests.txt 2>&1
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-sampling --no-roofline"
( MDM_HIP_ATTN_BWD=small16 timeout 200 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('attn16 (L=1024 old split too)', d['ms_per_step'])" ) > gpurun_out/c4_bench.txt 2>&1
( timeout 200 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('attn32', d['ms_per_step'])" ) >> gpurun_out/c4_bench.txt 2>&1
( MDM_HIP_ATTN_BWD=small16 timeout 200 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('attn16 again', d['ms_per_step'])" ) >> gpurun_out/c4_bench.txt 2>&1
( timeout 200 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('attn32 again', d['ms_per_step'])" ) >> gpurun_out/c4_bench.txt 2>&1
grep -v "^B=" gpurun_out/c4_attn_debug.txt | tail -5; grep -c " ok" gpurun_out/c4_attn_debug.txt; grep "FAIL" gpurun_out/c4_attn_debug.txt | head; cat gpurun_out/c4_stamps.txt; tail -3 gpurun_out/c4_attn_tests.txt; cat gpurun_out/c4_bench.txt
