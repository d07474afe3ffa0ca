#!/bin/bash
# the 1x1 shapes of the 16x16 level under each forced tile (fresh-buffer ring, all epilogue kinds)
cd /root/repo
export TMPDIR=/tmp
for t in 0 256256 256192 128128; do
echo "== KB_TILE=$t"
KB_TILE=$t timeout 300 python tools/kbench.py rotate 2>&1 | grep "768->3072\|3072->768\|768->2304\|768->768\|512->2048" | cut -c1-100
done
