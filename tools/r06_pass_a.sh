#!/bin/bash
# Round-6 pass a (baseline of the round-5 tree on this round's box): headline bench, overlapped trace -> critical path,
# and the first rocprofv3 passes over the 1024(W)x320(H) secondary workload.   usage: tools/r06_pass_a.sh [outdir-name]
R=${1:-r06a}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd $ROOT
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/bench_quick.json 2> $OUT/bench_quick.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT/prof2 -o kt -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary > $OUT/bench_under_rocprof_overlapped.log 2>&1
JP_POSE_STREAM=0 JP_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_s -o kt -- python $ROOT/bench.py --secondary-only --steps 5 --warmup 2 > $OUT/sec_under_rocprof.log 2>&1
timeout 600 rocprofv3 --kernel-trace -d $OUT/prof_s2 -o kt -- python $ROOT/bench.py --secondary-only --steps 5 --warmup 2 > $OUT/sec_under_rocprof_overlapped.log 2>&1
cd $ROOT
python tools/timeline.py $(find $OUT/prof2 -name "*.db" | head -1) 2 > $OUT/timeline.txt 2>&1
python tools/critical_path.py $(find $OUT/prof2 -name "*.db" | head -1) 2 > $OUT/critical_path.txt 2>&1
python tools/rocpd_stats.py $(find $OUT/prof_s -name "*.db" | head -1) 90 > $OUT/kernel_stats_320x1024.md 2>&1
python tools/timeline.py $(find $OUT/prof_s2 -name "*.db" | head -1) 2 > $OUT/timeline_320x1024.txt 2>&1
python tools/critical_path.py $(find $OUT/prof_s2 -name "*.db" | head -1) 2 > $OUT/critical_path_320x1024.txt 2>&1
rm -rf $OUT/prof2 $OUT/prof_s $OUT/prof_s2
cat $OUT/bench_quick.json | cut -c1-400; head -30 $OUT/critical_path.txt
