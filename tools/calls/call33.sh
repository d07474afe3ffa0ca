#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
echo "--- cold, dres only"; ( KB_GN_COLD=8 KB_GN_RES=1 timeout 300 python tools/kbench.py gn ) 2>&1 | grep "^gn 16\|rror"
echo "--- cold, FiLM + dres + dres2 (bytes counted as 4 passes: really 5)"; ( KB_GN_FULL=1 KB_GN_COLD=8 KB_GN_RES=1 timeout 300 python tools/kbench.py gn ) 2>&1 | grep "^gn 16\|rror"
echo "--- cold, no residual"; ( KB_GN_COLD=8 timeout 300 python tools/kbench.py gn ) 2>&1 | grep "^gn 16\|rror"
