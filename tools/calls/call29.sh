#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python tools/attn_debug.py time ) 2>&1 | grep "fwd\|ALL\|FAIL\|attn B" | tee gpurun_out/c29_attn_fwd32.txt | tail -30
