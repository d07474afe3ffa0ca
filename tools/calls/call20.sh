#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
for round in 1 2; do
for v in "" gelu_poly gelu_exact; do
  if [ -z "$v" ]; then echo "== sigmoid form (default)"; KB_RING=6 timeout 200 python tools/kbench.py rotate 2>&1 | grep "768->3072.*act=[12]\|512->2048.*act=[12]" ;
  else echo "== $v"; MDM_HIP_LIB=ml-mdm_amd/mdm_hip/lib_$v.so KB_RING=6 timeout 200 python tools/kbench.py rotate 2>&1 | grep "768->3072.*act=[12]\|512->2048.*act=[12]"; fi
done
done
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline --no-sampling"
for i in 1 2; do
( timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step sigmoid', d['ms_per_step'])" ) 2>/dev/null
( MDM_HIP_LIB=ml-mdm_amd/mdm_hip/lib_gelu_poly.so timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step poly', d['ms_per_step'])" ) 2>/dev/null
( MDM_HIP_LIB=ml-mdm_amd/mdm_hip/lib_gelu_exact.so timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step exact', d['ms_per_step'])" ) 2>/dev/null
done
