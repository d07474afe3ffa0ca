#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "--- gn new (cold)"; ( KB_GN_COLD=8 KB_GN_RES=1 timeout 300 python tools/kbench.py gn ) 2>&1 | grep "^gn\|Error\|error" | tee gpurun_out/c28_gn_new_cold.txt
echo "--- gn old (cold)"; ( MDM_HIP_LIB=/root/repo/ml-mdm_amd/mdm_hip/lib_oldgn.so KB_GN_COLD=8 KB_GN_RES=1 timeout 300 python tools/kbench.py gn ) 2>&1 | grep "^gn\|Error\|error" | tee gpurun_out/c28_gn_old_cold.txt
