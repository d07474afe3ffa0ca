#!/bin/bash
# round-2 GPU call 1: full GPU test suite, default bench, tile / stream experiments, rocprof of the bench
mkdir -p gpurun_out/r2c1; O=gpurun_out/r2c1
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -s --maxfail=15 -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench_default.log 2>&1
Q="--steps 8 --warmup 3 --no-cpu-baseline --no-sampling --no-nested"
MDM_HIP_TILE_EXP=1024 timeout 300 python bench.py $Q > $O/bench_tile1024.log 2>&1
MDM_HIP_TILE_EXP=4096 timeout 300 python bench.py $Q > $O/bench_tile4096.log 2>&1
MDM_HIP_WGRAD_STREAMS=2 timeout 300 python bench.py $Q --no-roofline > $O/bench_ws2.log 2>&1
MDM_HIP_WGRAD_STREAMS=4 timeout 300 python bench.py $Q --no-roofline > $O/bench_ws4.log 2>&1
MDM_HIP_SIDE_CUMASK=55555555 timeout 300 python bench.py $Q --no-roofline > $O/bench_mask55.log 2>&1
MDM_HIP_SIDE_CUMASK=0000ffff timeout 300 python bench.py $Q --no-roofline > $O/bench_mask0f.log 2>&1
for e in 0 4096; do
  for a in 0 1; do
    MDM_HIP_TILE_EXP=$e KB_ACT=$a timeout 200 python tools/kbench.py fwd > $O/kbench_fwd_exp${e}_act${a}.log 2>&1
  done
done
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof -o bench -- python /root/repo/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-nested > /root/repo/$O/prof.log 2>&1
cd /root/repo; ls $O/prof | head; python tools/kstats_db.py $(ls $O/prof/*.db | head -1) 70 > $O/kstats.txt 2>&1
tail -3 $O/tests.log; tail -c 600 $O/bench_default.log
