#!/bin/bash
# PMC counters of the standalone P9S / W9S harnesses in the three-product build: MFMA busy, waits, LDS, clock
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/${1:-r05ns2}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in ${2:-p9s_ns2_ah1 w9s_ns2_la1 p9s_ns3_k1 w9s_ns3}; do
  echo "== $v" >> $O/harness_pmc.log
  $ROOT/ubench_bin/$v 5 >> $O/harness_pmc.log 2>&1
  rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_$v -o p -- $ROOT/ubench_bin/$v 4 > $O/pmc_$v.log 2>&1
  python $ROOT/tools/pmc_dump.py $(find $O/pmc_$v -name "*.db" | head -1) 2>&1 | grep -B1 -A9 "GRBM_GUI_ACTIVE" >> $O/harness_pmc.log
  rm -rf $O/pmc_$v
  rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS -d $O/pmc2_$v -o p -- $ROOT/ubench_bin/$v 4 > $O/pmc2_$v.log 2>&1
  python $ROOT/tools/pmc_dump.py $(find $O/pmc2_$v -name "*.db" | head -1) 2>&1 | grep -B1 -A9 "SQ_ACTIVE_INST_LDS" >> $O/harness_pmc.log
  rm -rf $O/pmc2_$v
done
cat $O/harness_pmc.log
