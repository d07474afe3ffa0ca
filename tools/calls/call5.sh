#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python tools/attn_debug.py time ) > gpurun_out/c5_attn_debug.txt 2>&1
( timeout 120 python tools/attn_debug.py stamps ) > gpurun_out/c5_stamps.txt 2>&1
( timeout 400 python -m pytest tests/test_ops_gpu.py -k "test_attention" -q ) > gpurun_out/c5_attn_tests.txt 2>&1
grep -v "^B=" gpurun_out/c5_attn_debug.txt | tail -5; grep -c " ok" gpurun_out/c5_attn_debug.txt; grep "FAIL" gpurun_out/c5_attn_debug.txt | head; cat gpurun_out/c5_stamps.txt; tail -3 gpurun_out/c5_attn_tests.txt
