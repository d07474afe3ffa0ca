#!/bin/bash
# side-stream operands without Tensor.record_stream (kept alive by reference until the stream has passed them): A/B + gaps
cd /root/repo
mkdir -p gpurun_out/r6
export TMPDIR=/tmp
O=/root/repo/gpurun_out/r6
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested1024 --no-roofline --no-sampling"
for i in 1 2 3; do
for m in product record; do
( if [ $m = record ]; then export MDM_HIP_RECORD_STREAM=1; fi; timeout 300 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step variant=$m', d['ms_per_step'], 'nested256', d['nested256']['ms_per_step'])" ) 2>&1 | tail -1
done
done
cd /tmp
( timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_g -o bench -- python /root/repo/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-nested --no-reference-loop --no-nested1024 --no-sampling --no-roofline > /dev/null ) 2> /dev/null
cd /root/repo
DB=$(find $O/prof_g -name "*.db" | head -1)
python tools/fwd_gaps.py $DB --gaps > $O/stream_gaps_after.txt 2>&1
rm -rf $O/prof_g
tail -36 $O/stream_gaps_after.txt | cut -c1-200
timeout 900 python -m pytest tests/test_trainer_gpu.py tests/test_distributed_gpu.py -x -q -m gpu 2>&1 | tail -3
