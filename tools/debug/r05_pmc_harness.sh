#!/bin/bash
# PMC counters of the standalone iconv-forward harness variants (tools/ubench/p9us2_bench.hip): MFMA busy, waits, clock
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/${1:-r05d}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in ${2:-old new noa nobr noab nostg}; do
  echo "== $v" >> $O/harness_pmc.log
  $ROOT/ubench_bin/p9us2_$v 6 >> $O/harness_pmc.log 2>&1
  rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_$v -o p -- $ROOT/ubench_bin/p9us2_$v 4 > $O/pmc_$v.log 2>&1
  python $ROOT/tools/pmc_dump.py $(find $O/pmc_$v -name "*.db" | head -1) 2>&1 | grep -A12 "p9us" >> $O/harness_pmc.log
  rm -rf $O/pmc_$v
done
cat $O/harness_pmc.log
