#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r5
( timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "conv or ffn or gemm or linear" ) > gpurun_out/r5/gelu_ops_tests.txt 2>&1; tail -3 gpurun_out/r5/gelu_ops_tests.txt
( timeout 900 python -m pytest tests/test_model_gpu.py -x -q ) > gpurun_out/r5/gelu_model_tests.txt 2>&1; tail -3 gpurun_out/r5/gelu_model_tests.txt
for v in "" gelu_exact; do
  if [ -z "$v" ]; then echo "== fast gelu (default)"; KB_RING=6 timeout 200 python tools/kbench.py rotate 2>&1 | grep "768->3072.*act=[12]\|512->2048.*act=[12]" ;
  else echo "== $v"; MDM_HIP_LIB=ml-mdm_amd/mdm_hip/lib_$v.so KB_RING=6 timeout 200 python tools/kbench.py rotate 2>&1 | grep "768->3072.*act=[12]\|512->2048.*act=[12]"; fi
done
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline"
for i in 1 2; do
( timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step fast-gelu', d['ms_per_step'], 'sampling', d['sampling']['ms_per_denoise_step'])" ) 2>/dev/null
( MDM_HIP_LIB=ml-mdm_amd/mdm_hip/lib_gelu_exact.so timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step exact-gelu', d['ms_per_step'], 'sampling', d['sampling']['ms_per_denoise_step'])" ) 2>/dev/null
done
