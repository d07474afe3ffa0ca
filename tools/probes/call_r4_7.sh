#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/shape_profile.py unet64 > $O/shapes_unet64.txt 2>&1
timeout 600 python tools/shape_profile.py unet64 --serial > $O/shapes_unet64_serial.txt 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_a -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-nested --no-sampling --no-reference-loop --no-nested1024 --no-roofline > $GRAFT_REPO_ROOT/$O/prof_a.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/kstats.py $(ls $O/prof_a/*kernel_stats.csv | head -1) 9 60 > $O/kstats_a.txt 2>&1
find $O/prof_a -name "*.csv" -size +3M -delete
cat $O/shapes_unet64_serial.txt | head -70
echo; head -50 $O/kstats_a.txt
