#!/bin/bash
# same-box A/B: the round-5 tree (staged under .ab_r05/) against HEAD, alternating, 10 timed steps each
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06ab
mkdir -p $OUT
: > $OUT/ab.log
for i in 1 2 3; do
  (cd $ROOT/.ab_r05 && timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r05 tree  ', d['ms_per_step'], 'ms/step', d['value'], 'images/s')") >> $OUT/ab.log
  (cd $ROOT && timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary --no-exact-build 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('HEAD (r06)', d['ms_per_step'], 'ms/step', d['value'], 'images/s')") >> $OUT/ab.log
done
cat $OUT/ab.log
