#!/bin/bash
R=${1:-r06d}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd $ROOT
timeout 600 python tools/debug/second_step_fresh.py 512 2 Argo_both > $OUT/second_step_fresh.log 2>&1
timeout 600 python tools/debug/step_repro.py 256 2 3 static > $OUT/step_repro.log 2>&1
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "scale" > $OUT/pytest_scale.log 2>&1
tail -20 $OUT/second_step_fresh.log; tail -12 $OUT/step_repro.log; tail -3 $OUT/pytest_scale.log
