#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4g
mkdir -p $O
export PYTHONPATH=$R/ml-mdm_amd
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_s -o samp -- python $R/tools/sample_bench.py unet64 4 24 > $O/prof_s.log 2>&1
DB=$(find $O/prof_s -name "*.db" | head -1)
python $R/tools/kstats_db.py $DB 45 > $O/sampling_unet64_b4_kernel_stats.txt 2>&1
rm -rf $O/prof_s
head -50 $O/sampling_unet64_b4_kernel_stats.txt | cut -c1-150
grep '^{' $O/prof_s.log | cut -c1-300
