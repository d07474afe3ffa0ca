#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python tools/attn_debug.py time ) > gpurun_out/c1_attn_debug.txt 2>&1
( timeout 400 python -m pytest tests/test_ops_gpu.py -k "test_attention" -x -q ) > gpurun_out/c1_attn_tests.txt 2>&1
( timeout 300 python -m pytest tests/test_trainer_gpu.py tests/test_model_gpu.py -k "trainer or non_finite or train" -x -q ) > gpurun_out/c1_trainer_tests.txt 2>&1
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-sampling --no-roofline"
( MDM_HIP_EARLY_LOSS_SYNC=1 MDM_HIP_ATTN_BWD=small16 timeout 200 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('early-sync + attn16', d['ms_per_step'])" ) > gpurun_out/c1_bench.txt 2>&1
( MDM_HIP_ATTN_BWD=small16 timeout 200 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('late-sync + attn16', d['ms_per_step'])" ) >> gpurun_out/c1_bench.txt 2>&1
( timeout 200 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('late-sync + attn32', d['ms_per_step'])" ) >> gpurun_out/c1_bench.txt 2>&1
tail -40 gpurun_out/c1_attn_debug.txt; tail -5 gpurun_out/c1_attn_tests.txt; tail -5 gpurun_out/c1_trainer_tests.txt; cat gpurun_out/c1_bench.txt
