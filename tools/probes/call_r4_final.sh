#!/bin/bash
# Round-4 measurement pass: GPU tests + smoke, the default bench line, rocprofv3 kernel trace (+ serial weight gradients),
# per-shape tables, the two HBM PMC passes.  Outputs under gpurun_out/r4f/.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4f
mkdir -p $O
export PYTHONPATH=$R/ml-mdm_amd
export TMPDIR=/tmp
if [ "$1" != "notests" ]; then
timeout 1000 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -3 $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
fi
timeout 900 python bench.py > $O/bench_default.log 2>&1; grep '^{' $O/bench_default.log > $O/bench_line.json; cut -c1-400 $O/bench_line.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-nested --no-reference-loop --no-nested1024 --no-sampling > $O/prof.log 2>&1
grep '^{' $O/prof.log > $O/bench_line_profiled.json
DB=$(find $O/prof -name "*.db" | head -1)
python $R/tools/kstats_db.py $DB 60 --train-steps > $O/kernel_stats.txt 2>&1
python $R/tools/fwd_gaps.py $DB > $O/stream_windows.txt 2>&1
head -12 $O/kernel_stats.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_serial -o bench -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-nested --no-reference-loop --no-nested1024 --no-sampling --no-roofline --serial-wgrad > $O/prof_serial.log 2>&1
grep '^{' $O/prof_serial.log > $O/bench_line_serial.json
DBS=$(find $O/prof_serial -name "*.db" | head -1)
python $R/tools/kstats_db.py $DBS 60 --train-steps > $O/kernel_stats_serial.txt 2>&1
cd $R
timeout 300 python tools/shape_profile.py unet64 > $O/shapes_unet64.txt 2>&1
timeout 300 python tools/shape_profile.py unet64 --serial > $O/shapes_unet64_serial.txt 2>&1
timeout 300 python tools/shape_profile.py nested256 > $O/shapes_nested256.txt 2>&1
cd /tmp
PMCCMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-nested --no-reference-loop --no-nested1024 --no-sampling --no-roofline"
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o pmc -- $PMCCMD > $O/pmc_f.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o pmc -- $PMCCMD > $O/pmc_w.log 2>&1
cd $R
python tools/pmc_traffic.py $O/pmc_f $O/pmc_w $O/pmc_hbm_traffic.json > $O/pmc_traffic.log 2>&1; tail -3 $O/pmc_traffic.log
rm -rf $O/pmc_f $O/pmc_w $O/prof $O/prof_serial
ls -la $O
