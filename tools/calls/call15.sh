#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== B=16 (B x H = 128 heads on 256 CUs)"; KB_BATCH=16 timeout 120 python tools/attn_debug.py timeonly 2>&1 | grep "^attn"
echo "== B=32"; KB_BATCH=32 timeout 120 python tools/attn_debug.py timeonly 2>&1 | grep "^attn"
echo "== B=4"; KB_BATCH=4 timeout 120 python tools/attn_debug.py timeonly 2>&1 | grep "^attn"
