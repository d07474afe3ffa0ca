#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python tools/attn_debug.py time ) > gpurun_out/c9_attn_debug.txt 2>&1
( MDM_HIP_LIB=ml-mdm_amd/mdm_hip/lib_nopipe.so timeout 120 python tools/attn_debug.py timeonly ) >> gpurun_out/c9_attn_debug.txt 2>&1
( timeout 120 python tools/attn_debug.py timeonly ) >> gpurun_out/c9_attn_debug.txt 2>&1
( timeout 400 python -m pytest tests/test_ops_gpu.py -k "test_attention" -q ) > gpurun_out/c9_attn_tests.txt 2>&1
grep -v "^B=" gpurun_out/c9_attn_debug.txt | tail -12; grep -c " ok" gpurun_out/c9_attn_debug.txt; grep "FAIL" gpurun_out/c9_attn_debug.txt | head; tail -3 gpurun_out/c9_attn_tests.txt
