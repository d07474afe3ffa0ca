#!/bin/bash
# AdamW + EMA with non-temporal loads (1), stores (2), both (3) against the plain kernel
cd /root/repo
export TMPDIR=/tmp
for r in 1 2; do
for m in product nt1 nt2 nt3; do
( if [ $m != product ]; then export MDM_HIP_LIB=/root/repo/ab_libs/lib_$m.so; fi; timeout 200 python tools/adam_bench.py 2>&1 | tail -1 )
done
done
