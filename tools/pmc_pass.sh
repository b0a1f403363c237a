#!/bin/bash
# GPU-box aid: only the two PMC passes of measure_round.sh (HBM traffic of the by-time dominant kernel).
# usage: tools/pmc_pass.sh r02   (needs gpurun_out/rNN/bench_families.json or profiles/rNN_bench_families.json)
R=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
FAM=$OUT/bench_families.json; [ -f $FAM ] || FAM=$ROOT/profiles/${R}_bench_families.json
cd /tmp && export TMPDIR=/tmp
JP_PMC_CALIB=1 timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc -o fetch -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary > $OUT/pmc_fetch.log 2>&1
JP_PMC_CALIB=1 timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc -o write -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary > $OUT/pmc_write.log 2>&1
cd $ROOT
python tools/pmc_traffic.py $(find $OUT/pmc -name "fetch*.db" | head -1) $(find $OUT/pmc -name "write*.db" | head -1) $OUT/pmc_traffic.json $FAM > $OUT/pmc_traffic.log 2>&1
rm -rf $OUT/pmc
tail -12 $OUT/pmc_traffic.log
