#!/bin/bash
# round 6, call 2: byte-coded gelu' (FFN): parity tests, then the step A/B against the variant that keeps the bf16 pre-activation
cd /root/repo
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "epilogue or gelu or ffn or conv or split" ) 2>&1 | tail -4
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline --no-sampling"
for i in 1 2 3; do
for m in byte bf16; do
( if [ $m = bf16 ]; then export MDM_HIP_LIB=/root/repo/ab_libs/lib_auxbf16.so; fi; timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step aux=$m', d['ms_per_step'])" ) 2>&1 | tail -1
done
done
