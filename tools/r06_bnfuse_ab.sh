#!/bin/bash
# same-box pairs: BatchNorm statistics from the convolutions' epilogues (default) vs the pass over y (JP_BN_STATS_FUSE=0), B = 8, 1024^2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06n
mkdir -p $OUT
cd $ROOT
: > $OUT/bn_stats_fuse_ab.log
for i in 1 2 3 4; do
  JP_BN_STATS_FUSE=0 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-exact-build --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('statistics pass over y      ', d['ms_per_step'], 'ms/step')" >> $OUT/bn_stats_fuse_ab.log
  python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-exact-build --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('statistics from the epilogue', d['ms_per_step'], 'ms/step')" >> $OUT/bn_stats_fuse_ab.log
done
cat $OUT/bn_stats_fuse_ab.log
