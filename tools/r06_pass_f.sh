#!/bin/bash
R=${1:-r06f}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd $ROOT
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-exact-build > $OUT/bench_quick.json 2> $OUT/bench_quick.err
JP_AMAX_LOG=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-exact-build > $OUT/amax_log.txt 2>&1
timeout 3000 python -m pytest tests -m gpu -q -x --deselect tests/test_split_accuracy_gpu.py > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -12 $OUT/pytest_gpu.log | cut -c1-250; cut -c1-200 $OUT/bench_quick.json; grep 'amax log' $OUT/amax_log.txt | head -24
