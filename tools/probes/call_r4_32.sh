#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4h
mkdir -p $O
export PYTHONPATH=ml-mdm_amd
timeout 1000 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -3 $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.log 2>&1; grep '^{' $O/bench_default.log > $O/bench_line.json; cut -c1-330 $O/bench_line.json
L=$O/sampling_latency.jsonl
rm -f $L
timeout 300 python tools/sample_bench.py nested1024 4 8 2>&1 | grep '^{' >> $L
timeout 300 python tools/sample_bench.py unet64 4 8 2>&1 | grep '^{' >> $L
timeout 300 python tools/sample_bench.py unet64 1 8 2>&1 | grep '^{' >> $L
timeout 300 python tools/sample_bench.py nested256 16 8 2>&1 | grep '^{' >> $L
cat $L
