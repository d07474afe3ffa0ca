#!/bin/bash
# idle time inside the graphed sampling iterations (one hipGraph replay per denoise iteration)
cd /root/repo
mkdir -p gpurun_out/r6
export TMPDIR=/tmp
O=/root/repo/gpurun_out/r6
for w in "nested1024 4" "unet64 64" "unet64 4"; do
set -- $w
cd /tmp
( timeout 500 rocprofv3 --kernel-trace -d $O/prof_s -o s -- python /root/repo/tools/sample_bench.py $1 $2 8 > $O/sample_$1_$2.json ) 2> /dev/null
cd /root/repo
DB=$(find $O/prof_s -name "*.db" | head -1)
echo "== $1 batch $2: $(tail -1 $O/sample_$1_$2.json | cut -c1-200)"
python tools/graph_gaps.py $DB 8
rm -rf $O/prof_s
done
