#!/bin/bash
# same-box step pairs: default vs one switch off (usage: r05_step_ab.sh <outdir> <ENVVAR=value> [pairs])
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=$ROOT/gpurun_out/${1:-r05s}; mkdir -p $O
SW=${2:-JP_P1L=0}; N=${3:-3}
for rep in $(seq 1 $N); do
  env $SW timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > $O/step_off_$rep.json 2>> $O/step.err
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > $O/step_on_$rep.json 2>> $O/step.err
done
for f in $O/step_*.json; do python -c "
import json
d=json.load(open('$f')); print('$SW', '$f'.split('/')[-1], d['ms_per_step'], d['value'])"; done | tee $O/step_ab.log
