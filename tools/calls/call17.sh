#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
( timeout 400 python -m pytest tests/test_ops_gpu.py -k "test_attention or profile or last_gemm" -q ) > gpurun_out/r5/attn_tests.txt 2>&1; tail -3 gpurun_out/r5/attn_tests.txt
( timeout 120 python tools/attn_debug.py timeonly ) 2>&1 | grep "^attn"
( timeout 120 python tools/attn_debug.py stamps ) 2>&1 | head -16
bash tools/calls/call16.sh
