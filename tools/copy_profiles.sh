#!/bin/bash
# copy the summaries of a measurement pass (tools/measure_round.sh, merged back under gpurun_out/<pass>/) into profiles/<round>_*
# usage: tools/copy_profiles.sh r03f r03
P=${1:?pass dir under gpurun_out}; R=${2:?round prefix}
S=$(dirname "$0")/../gpurun_out/$P; D=$(dirname "$0")/../profiles
cp $S/bench_n1.json $D/${R}_bench_n1.json
cp $S/bench_families.json $D/${R}_bench_families.json
cp $S/kernel_stats.md $D/${R}_kernel_stats_bench_b8_1024.md
cp $S/kernel_stats_overlapped.md $D/${R}_kernel_stats_overlapped.md
cp $S/timeline.txt $D/${R}_timeline.txt
cp $S/pmc_traffic.json $D/${R}_pmc_traffic.json
cp $S/pmc_traffic_kernels.md $D/${R}_pmc_traffic_kernels.md
cp $S/pytest_gpu.log $D/${R}_pytest_gpu.log
cp $S/smoke.log $D/${R}_smoke.log
cp $S/bench_under_rocprof.log $D/${R}_bench_under_rocprof.log
cp $S/bench_other_configs.jsonl $D/${R}_bench_other_configs.jsonl
ls -la $D | grep ${R}_
for f in kernel_stats_320x1024.md critical_path_traced.txt stream_milestones.md copybuffer_count.md amax_log.txt step_repro.log; do [ -f $S/$f ] && cp $S/$f $D/${R}_$f; done
ls -la $D | grep ${R}_
