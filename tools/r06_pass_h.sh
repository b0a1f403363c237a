#!/bin/bash
R=${1:-r06h}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "scale or pose or smooth" > $OUT/pytest_scale.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_scale.log
timeout 600 python tools/debug/step_repro.py 256 2 3 static DepthEncoder > $OUT/step_repro.log 2>&1
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/bench_quick.json 2> $OUT/bench_quick.err
cd /tmp && export TMPDIR=/tmp
JP_POSE_STREAM=0 JP_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_s -o kt -- python $ROOT/bench.py --secondary-only --steps 5 --warmup 2 > $OUT/sec_under_rocprof.log 2>&1
cd $ROOT
python tools/rocpd_stats.py $(find $OUT/prof_s -name "*.db" | head -1) 90 > $OUT/kernel_stats_320x1024.md 2>&1
rm -rf $OUT/prof_s
tail -3 $OUT/pytest_scale.log; grep -v "Exception\|Traceback\|ops.py\|Attribute\|warn\|Warn\|resnet" $OUT/step_repro.log | tail -70; python -c "
import json; d=json.load(open('$OUT/bench_quick.json')); print(d['value'], d['ms_per_step'], d['secondary']['value'], d['exact_build'])"; grep scale_bwd $OUT/kernel_stats_320x1024.md | cut -c1-150
