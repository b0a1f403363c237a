#!/bin/bash
# overlapped trace of the headline step -> critical path with the chain in time order (tools/critical_path.py)
R=${1:-r06b}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT/prof2 -o kt -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary > $OUT/bench_under_rocprof_overlapped.log 2>&1
cd $ROOT
python tools/critical_path.py $(find $OUT/prof2 -name "*.db" | head -1) 2 > $OUT/critical_path.txt 2>&1
python tools/critical_path.py $(find $OUT/prof2 -name "*.db" | head -1) 3 > $OUT/critical_path_step3.txt 2>&1
rm -rf $OUT/prof2
