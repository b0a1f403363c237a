#!/bin/bash
R=${1:-r06e}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd $ROOT
timeout 2400 python -m pytest tests/test_multi_step_parity_gpu.py tests/test_split_accuracy_gpu.py tests/test_amax_gpu.py tests/test_kernels_gpu.py tests/test_step_parity_gpu.py -q -x > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_subset.log
timeout 600 python tools/debug/step_repro.py 256 2 3 static > $OUT/step_repro.log 2>&1
timeout 900 python tools/operand_ranges.py --train 40 > $OUT/operand_ranges.md 2> $OUT/operand_ranges.err
tail -8 $OUT/pytest_subset.log; grep -v "Exception\|Traceback\|ops.py\|Attribute\|warn\|Warn\|resnet" $OUT/step_repro.log | head -12; head -20 $OUT/operand_ranges.md
