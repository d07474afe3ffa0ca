#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "norm" ) 2>&1 | tail -4
echo "--- gn new"; ( KB_GN_RES=1 timeout 200 python tools/kbench.py gn ) 2>&1 | grep "^gn" | tee gpurun_out/c27_gn_new.txt
echo "--- gn old"; ( MDM_HIP_LIB=/root/repo/ml-mdm_amd/mdm_hip/lib_oldgn.so KB_GN_RES=1 timeout 200 python tools/kbench.py gn ) 2>&1 | grep "^gn" | tee gpurun_out/c27_gn_old.txt
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline --no-sampling"
for i in 1 2; do
( timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step new', d['ms_per_step'])" ) 2>/dev/null
( MDM_HIP_LIB=ml-mdm_amd/mdm_hip/lib_oldgn.so timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step old gn', d['ms_per_step'])" ) 2>/dev/null
done
