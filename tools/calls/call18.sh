#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
for v in "" nogelu "" nogelu; do
  if [ -z "$v" ]; then echo "== default"; KB_RING=6 timeout 200 python tools/kbench.py rotate 2>&1 | grep "768->3072\|512->2048" ;
  else echo "== $v"; MDM_HIP_LIB=ml-mdm_amd/mdm_hip/lib_$v.so KB_RING=6 timeout 200 python tools/kbench.py rotate 2>&1 | grep "768->3072\|512->2048"; fi
done
B="--steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-sampling --no-roofline"
( timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step default', d['ms_per_step'])" ) 2>/dev/null
( MDM_HIP_LIB=ml-mdm_amd/mdm_hip/lib_nogelu.so timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step nogelu (wrong numerics)', d['ms_per_step'])" ) 2>/dev/null
