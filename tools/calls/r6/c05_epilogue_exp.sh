#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
for k in 0 1 2 3 4 6; do
echo "--- knob3=$k (bit0: no code store, bit1: no encode arithmetic, bit2: + the 16-byte pre-activation store)"
( KB_KNOB3=$k timeout 250 python tools/kbench.py rotate 2>&1 | grep "act=1" | grep "3072\|2048" )
done
