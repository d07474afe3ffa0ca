#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline --no-sampling"
for i in 1 2; do
( timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step default', d['ms_per_step'])" ) 2>/dev/null
( MDM_HIP_LIB=ml-mdm_amd/mdm_hip/lib_nosilu.so timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step nosilu (wrong numerics)', d['ms_per_step'])" ) 2>/dev/null
done
