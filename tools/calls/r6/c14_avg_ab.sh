#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline --no-sampling"
for i in 1 2; do
( timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step plain', d['ms_per_step'])" ) 2>&1 | tail -1
( timeout 250 python bench.py $B --force-collectives --bucket-mb 64 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step forced-collectives AVG', d['ms_per_step'])" ) 2>&1 | tail -1
( MDM_HIP_NO_AVG=1 timeout 250 python bench.py $B --force-collectives --bucket-mb 64 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step forced-collectives SUM', d['ms_per_step'])" ) 2>&1 | tail -1
( MDM_HIP_LIB=/root/repo/ab_libs/lib_r5gelu.so MDM_HIP_NO_AVG=1 timeout 250 python bench.py $B --force-collectives --bucket-mb 64 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step forced-collectives SUM r5lib', d['ms_per_step'])" ) 2>&1 | tail -1
( MDM_HIP_LIB=/root/repo/ab_libs/lib_r5gelu.so timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step plain r5lib', d['ms_per_step'])" ) 2>&1 | tail -1
done
