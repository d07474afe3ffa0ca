#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
for mb in 160 96 64 32; do
echo "--- chunk threshold $mb MiB"; ( MDM_HIP_GN_CHUNK_MB=$mb KB_GN_COLD=8 KB_GN_RES=1 timeout 200 python tools/kbench.py gn ) 2>&1 | grep "^gn 64\|^gn 32x32 C=768\|^gn 32x32 C=1280" | sed 's/(cold.*//' | awk '{print $2, $3, "bwd", $(NF-4), $(NF-3)}'
done
