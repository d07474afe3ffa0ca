#!/bin/bash
# round 6, call 26: LDS-tiled weight packs of the resampling convolutions, vectorised multi-LayerNorm kernels: parity + step A/B
cd /root/repo
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "text_kv or upsample or conv_fwd_bwd or layer_norm or s2 or shared_input" ) 2>&1 | tail -2
( timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "not long_horizon" ) 2>&1 | tail -2
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline --no-sampling"
for i in 1 2 3; do
for m in product prev; do
( if [ $m != product ]; then export MDM_HIP_LIB=/root/repo/ab_libs/lib_$m.so; fi; timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step variant=$m', d['ms_per_step'])" ) 2>&1 | tail -1
done
done
