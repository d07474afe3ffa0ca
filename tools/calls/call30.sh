#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "norm or attention" ) 2>&1 | tail -3
echo "--- gn new (cold)"; ( KB_GN_COLD=8 KB_GN_RES=1 timeout 300 python tools/kbench.py gn ) 2>&1 | grep "^gn 16\|^gn 32x32 C=512" | tee gpurun_out/c30_gn_new_cold.txt
echo "--- gn no prefetch (cold)"; ( MDM_HIP_LIB=/root/repo/ml-mdm_amd/mdm_hip/lib_nopre.so KB_GN_COLD=8 KB_GN_RES=1 timeout 300 python tools/kbench.py gn ) 2>&1 | grep "^gn 16\|^gn 32x32 C=512" | tee gpurun_out/c30_gn_nopre_cold.txt
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline --no-sampling"
for i in 1 2; do
( timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step new', d['ms_per_step'])" ) 2>/dev/null
( MDM_HIP_LIB=ml-mdm_amd/mdm_hip/lib_nopre.so timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step no dres prefetch', d['ms_per_step'])" ) 2>/dev/null
( MDM_HIP_SKIP_WGRAD_REDUCE=1 timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step no wgrad reduces (ablation)', d['ms_per_step'])" ) 2>/dev/null
done
