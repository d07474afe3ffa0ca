#!/bin/bash
# round 6, call 6: the epilogue's store loop without the per-chunk vector load of the development knob (+ its vmcnt(0)):
# per-kernel timing on fresh buffers (three builds), parity of the GEMM tests, step A/B
cd /root/repo
export TMPDIR=/tmp
for m in byte bf16pk r5; do
echo "--- variant=$m"
( if [ $m = bf16pk ]; then export MDM_HIP_LIB=/root/repo/ab_libs/lib_auxbf16.so; fi; if [ $m = r5 ]; then export MDM_HIP_LIB=/root/repo/ab_libs/lib_r5gelu.so; fi; timeout 250 python tools/kbench.py rotate 2>&1 | grep "fresh" )
done
( timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "epilogue or gelu or ffn or conv or split or linear" ) 2>&1 | tail -3
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline --no-sampling"
for i in 1 2 3; do
for m in byte bf16pk r5; do
( if [ $m = bf16pk ]; then export MDM_HIP_LIB=/root/repo/ab_libs/lib_auxbf16.so; fi; if [ $m = r5 ]; then export MDM_HIP_LIB=/root/repo/ab_libs/lib_r5gelu.so; fi; timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step variant=$m', d['ms_per_step'])" ) 2>&1 | tail -1
done
done
