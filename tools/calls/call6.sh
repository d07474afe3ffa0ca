#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in "" sr64p0 sr128p0 sr128p2 "" sr64p0; do
  if [ -z "$v" ]; then echo "== default (sr64 prio-flip)"; timeout 120 python tools/attn_debug.py timeonly 2>&1 | grep "^attn";
  else echo "== $v"; MDM_HIP_LIB=ml-mdm_amd/mdm_hip/lib_$v.so timeout 120 python tools/attn_debug.py timeonly 2>&1 | grep "^attn"; fi
done > gpurun_out/c6_variants.txt 2>&1
cat gpurun_out/c6_variants.txt
