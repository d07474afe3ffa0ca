#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python tools/attn_debug.py time ) > gpurun_out/c8_attn_debug.txt 2>&1
( timeout 120 python tools/attn_debug.py timeonly ) >> gpurun_out/c8_attn_debug.txt 2>&1
grep -v "^B=" gpurun_out/c8_attn_debug.txt | tail -8; grep -c " ok" gpurun_out/c8_attn_debug.txt; grep "FAIL" gpurun_out/c8_attn_debug.txt | head
