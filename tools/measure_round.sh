#!/bin/bash
# GPU-box aid: the round's measurement pass -> gpurun_out/rNN/ (copy the summaries into profiles/ afterwards).
# usage: tools/measure_round.sh r02
R=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd $ROOT
export JP_BENCH_PREWARM=0      # the profiled / short bench runs below count on steps + warmup iterations exactly; the headline line pre-warms
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
JP_BENCH_PREWARM=30 JP_BENCH_TABLE=$OUT/bench_families.json timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
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
# 1024(W) x 320(H) secondary workload: single-stream kernel stats (round 6)
JP_POSE_STREAM=0 JP_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_s -o kt -- python $ROOT/bench.py --secondary-only --steps 5 --warmup 2 > $OUT/sec_under_rocprof.log 2>&1
# set-up copies vs per-step copies: the same trace at 2 and at 10 steps
timeout 600 rocprofv3 --kernel-trace -d $OUT/prof_c2 -o kt -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-exact-build > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace -d $OUT/prof_c10 -o kt -- python $ROOT/bench.py --steps 10 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-exact-build > /dev/null 2>&1
cd $ROOT
python - > $OUT/copybuffer_count.md 2>&1 <<PY
import sqlite3, glob
print("# __amd_rocclr_copyBuffer dispatches of bench.py (configs[1], B = 8, 1024^2) under rocprofv3 --kernel-trace")
for tag, steps in (("prof_c2", 3), ("prof_c10", 11)):
    db = glob.glob("$OUT/" + tag + "/**/*.db", recursive=True)[0]
    c = sqlite3.connect(db)
    n = c.execute("select count(*) from kernels where name like '%copyBuffer%'").fetchone()[0]
    k = c.execute("select count(*) from kernels").fetchone()[0]
    print(f"* {steps} training steps (warm-up included): {n} copyBuffer dispatches of {k} kernel dispatches")
PY
python tools/rocpd_stats.py $(find $OUT/prof_s -name "*.db" | head -1) 90 > $OUT/kernel_stats_320x1024.md 2>&1
python tools/critical_path.py $(find $OUT/prof2 -name "*.db" | head -1) 2 > $OUT/critical_path_traced.txt 2>&1
timeout 300 python tools/stream_milestones.py > $OUT/stream_milestones.md 2> /dev/null
JP_AMAX_LOG=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-exact-build 2>/dev/null | grep 'amax log' > $OUT/amax_log.txt
timeout 600 python tools/debug/step_repro.py 1024 8 2 static 2>/dev/null | grep -v "Exception\|Traceback\|ops.py\|Attribute" > $OUT/step_repro.log
python tools/rocpd_stats.py $(find $OUT/prof -name "*.db" | head -1) 80 > $OUT/kernel_stats.md 2>&1
python tools/rocpd_stats.py $(find $OUT/prof2 -name "*.db" | head -1) 80 > $OUT/kernel_stats_overlapped.md 2>&1
python tools/timeline.py $(find $OUT/prof2 -name "*.db" | head -1) 2 > $OUT/timeline.txt 2>&1
python tools/pmc_traffic.py $(find $OUT/pmc -name "fetch*.db" | head -1) $(find $OUT/pmc -name "write*.db" | head -1) $OUT/pmc_traffic.json $OUT/bench_families.json > $OUT/pmc_traffic.log 2>&1
rm -rf $OUT/prof $OUT/prof2 $OUT/pmc $OUT/prof_s $OUT/prof_c2 $OUT/prof_c10   # the databases are large; the summaries above are what gets committed
tail -3 $OUT/pytest_gpu.log; cat $OUT/smoke.log | tail -1; cat $OUT/bench_n1.json | cut -c1-1500; head -14 $OUT/kernel_stats.md; tail -8 $OUT/pmc_traffic.log
