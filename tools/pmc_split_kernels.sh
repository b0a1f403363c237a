#!/bin/bash
# GPU-box aid: MFMA-instruction / MFMA-busy / wait counters of the split-bf16 kernels at the step's largest shapes
# (rocprofv3 --pmc, counters only; one pass per process) -> gpurun_out/<pass>/pmc_split_kernels.txt
# usage: tools/pmc_split_kernels.sh r05p
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/${1:-r05p}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
C="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
: > $O/pmc_split_kernels.txt
run() {   # tag, command...
  local tag=$1; shift
  echo "#### $tag (un-profiled times first)" >> $O/pmc_split_kernels.txt
  "$@" 2>&1 | grep -E "ms/step|p9us|==" | grep -v Border >> $O/pmc_split_kernels.txt
  rocprofv3 --pmc $C -d $O/pmc_$tag -o p -- "$@" > $O/pmc_$tag.log 2>&1
  python $ROOT/tools/pmc_dump.py $(find $O/pmc_$tag -name "*.db" | head -1) 2>&1 | grep -A9 -E "p9us2|p9s_wide_kernel<4, 2|wgrad_w9s_kernel<4|p9s_kernel<4, 2, 2, false, false|w1s" >> $O/pmc_split_kernels.txt
  rm -rf $O/pmc_$tag
}
run conv3x3 python $ROOT/tools/conv_bench.py --iters 3 --only "merge 256->256 3x3 refl @256"
run conv1x1 python $ROOT/tools/conv_bench.py --iters 3 --only "CRP 256->256 1x1 @256"
run iconv python $ROOT/tools/debug/p9us_time.py
cat $O/pmc_split_kernels.txt
