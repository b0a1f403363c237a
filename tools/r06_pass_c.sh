#!/bin/bash
# full GPU test suite + quick bench + un-traced stream milestones + operand-magnitude reductions still made as separate passes
R=${1:-r06c}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd $ROOT
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-exact-build > $OUT/bench_quick.json 2> $OUT/bench_quick.err
timeout 300 python tools/stream_milestones.py > $OUT/stream_milestones.md 2> $OUT/stream_milestones.err
JP_AMAX_LOG=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-exact-build > $OUT/amax_log.txt 2>&1
timeout 3000 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -25 $OUT/pytest_gpu.log; cut -c1-300 $OUT/bench_quick.json; grep 'amax log' $OUT/amax_log.txt | head -30; cat $OUT/stream_milestones.md
