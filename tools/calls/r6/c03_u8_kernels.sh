#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -k "test_conv_gemm_epilogues" ) 2>&1 | grep -v "^$" | tail -40
for m in byte bf16; do
echo "--- aux=$m"
( if [ $m = bf16 ]; then export MDM_HIP_LIB=/root/repo/ab_libs/lib_auxbf16.so; fi; timeout 250 python tools/kbench.py rotate 2>&1 | grep "act=1\|act=2" )
done
