#!/bin/bash
# GPU-box aid: the round's measurement pass -> gpurun_out/rNN/ (copy the summaries into profiles/ afterwards).
# usage: tools/measure_round.sh r02
R=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd $ROOT
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
JP_BENCH_TABLE=$OUT/bench_families.json timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
# the other BASELINE.json configs on one GPU (config 0 / 4 = one image per GPU: with and without the captured step)
: > $OUT/bench_other_configs.jsonl
for c in 0 4; do for g in off on; do timeout 300 python bench.py --config $c --graph $g --steps 20 --warmup 4 --no-cpu-baseline --no-secondary --no-roofline >> $OUT/bench_other_configs.jsonl 2>> $OUT/bench_other.err; done; done
for c in 2 3; do timeout 600 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline >> $OUT/bench_other_configs.jsonl 2>> $OUT/bench_other.err; done
cd /tmp && export TMPDIR=/tmp
# (a) single-stream pass (no side / companion streams): per-kernel durations that are the kernels' own, comparable with
#     bench.py's HIP-event roofline;  (b) the default overlapped step: timeline (kernels in flight, idle gaps)
JP_POSE_STREAM=0 JP_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o kt -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary > $OUT/bench_under_rocprof.log 2>&1
timeout 600 rocprofv3 --kernel-trace -d $OUT/prof2 -o kt -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary > $OUT/bench_under_rocprof_overlapped.log 2>&1
JP_PMC_CALIB=1 timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc -o fetch -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary > $OUT/pmc_fetch.log 2>&1
JP_PMC_CALIB=1 timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc -o write -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary > $OUT/pmc_write.log 2>&1
cd $ROOT
python tools/rocpd_stats.py $(find $OUT/prof -name "*.db" | head -1) 80 > $OUT/kernel_stats.md 2>&1
python tools/rocpd_stats.py $(find $OUT/prof2 -name "*.db" | head -1) 80 > $OUT/kernel_stats_overlapped.md 2>&1
python tools/timeline.py $(find $OUT/prof2 -name "*.db" | head -1) 2 > $OUT/timeline.txt 2>&1
python tools/pmc_traffic.py $(find $OUT/pmc -name "fetch*.db" | head -1) $(find $OUT/pmc -name "write*.db" | head -1) $OUT/pmc_traffic.json $OUT/bench_families.json > $OUT/pmc_traffic.log 2>&1
rm -rf $OUT/prof $OUT/prof2 $OUT/pmc   # the databases are large; the summaries above are what gets committed
tail -3 $OUT/pytest_gpu.log; cat $OUT/smoke.log | tail -1; cat $OUT/bench_n1.json | cut -c1-1500; head -14 $OUT/kernel_stats.md; tail -8 $OUT/pmc_traffic.log
