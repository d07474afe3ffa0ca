#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( KB_DBIAS=1 timeout 200 python tools/kbench.py wgrad ) 2>&1 | tee gpurun_out/c25_wgrad_dbias.txt
( timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "conv or linear or wgrad or upconv or bias" ) 2>&1 | tail -4
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline --no-sampling"
for i in 1 2; do
( timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step new', d['ms_per_step'])" ) 2>/dev/null
( MDM_HIP_LIB=ml-mdm_amd/mdm_hip/lib_oldbias.so timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step old bias sums', d['ms_per_step'])" ) 2>/dev/null
done
