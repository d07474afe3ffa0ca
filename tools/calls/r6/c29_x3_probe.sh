#!/bin/bash
# fp32 (bf16x3) sampling: kernel breakdown + what pre-split operands could buy (timing-only variant libraries)
cd /root/repo
export TMPDIR=/tmp
for m in product x3b x3ab product x3b; do
( if [ $m != product ]; then export MDM_HIP_LIB=/root/repo/ab_libs/lib_$m.so; fi; timeout 300 python tools/sample_x3_probe.py 4 2>&1 | tail -1 )
done
cd /tmp
( timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof -o b -- python /root/repo/tools/sample_x3_probe.py 3 > /dev/null ) 2> /dev/null
cd /root/repo
DB=$(find /tmp/prof -name "*.db" | head -1)
python tools/kstats_db.py $DB 40 2>&1 | head -50
