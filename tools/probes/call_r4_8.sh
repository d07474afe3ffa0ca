#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sampling --no-reference-loop --no-nested1024 --no-roofline"
for r in 1 2; do
  for v in default nox; do
    if [ $v = nox ]; then export MDM_HIP_GEMM_X=0; else unset MDM_HIP_GEMM_X; fi
    timeout 300 $B 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['nested256']['ms_per_step'] if d.get('nested256') else None)"
  done
done
unset MDM_HIP_GEMM_X
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv_fwd_bwd" 2>&1 | tail -2
