#!/bin/bash
R=${1:-r06g}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_split_accuracy_gpu.py tests/test_amax_gpu.py tests/test_subpath_320x1024_gpu.py tests/test_bench_shapes_gpu.py -q > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_subset.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/bench_quick.json 2> $OUT/bench_quick.err
for c in 0 4; do for g in off on; do timeout 300 python bench.py --config $c --graph $g --steps 20 --warmup 4 --no-cpu-baseline --no-secondary --no-roofline --no-exact-build >> $OUT/bench_other_configs.jsonl 2>> $OUT/bench_other.err; done; done
cd /tmp && export TMPDIR=/tmp
JP_POSE_STREAM=0 JP_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_s -o kt -- python $ROOT/bench.py --secondary-only --steps 5 --warmup 2 > $OUT/sec_under_rocprof.log 2>&1
cd $ROOT
python tools/rocpd_stats.py $(find $OUT/prof_s -name "*.db" | head -1) 90 > $OUT/kernel_stats_320x1024.md 2>&1
rm -rf $OUT/prof_s
tail -8 $OUT/pytest_subset.log | cut -c1-200; python -c "
import json; d=json.load(open('$OUT/bench_quick.json')); print(d['value'], d['ms_per_step'], d['secondary']['value'], d['exact_build'])"; cut -c1-160 $OUT/bench_other_configs.jsonl; head -12 $OUT/kernel_stats_320x1024.md | cut -c1-180
