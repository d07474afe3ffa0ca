#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in "" abl1 abl2 abl4 abl8 abl12 abl16 abl32 abl63 ""; do
  if [ -z "$v" ]; then echo "== default"; timeout 120 python tools/attn_debug.py timeonly 2>&1 | grep "^attn" | sed 's/bwd\[split16\].*bwd\[small32\]/bwd[small32]/; s/bwd\[small16\] [0-9.]* ms [0-9]* TF//';
  else echo "== $v"; MDM_HIP_LIB=ml-mdm_amd/mdm_hip/lib_$v.so timeout 120 python tools/attn_debug.py timeonly 2>&1 | grep "^attn" | sed 's/bwd\[split16\].*bwd\[small32\]/bwd[small32]/; s/bwd\[small16\] [0-9.]* ms [0-9]* TF//'; fi
done > gpurun_out/c7_ablations.txt 2>&1
cat gpurun_out/c7_ablations.txt
