#!/bin/bash
# round-5 GPU session B: P9US2 (re-laid iconv forward stream) -- parity, kernel A/B, step trace, whole-step A/B
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=$ROOT/gpurun_out/${1:-r05b}; mkdir -p $O
C=$ROOT/jperceiver_amd/csrc
timeout 900 python -m pytest tests/test_bench_shapes_gpu.py -k "iconv" -x -q > $O/pytest_iconv.log 2>&1
timeout 900 python -m pytest tests/test_split_accuracy_gpu.py -x -q >> $O/pytest_iconv.log 2>&1
for rep in 1 2 3; do
  JP_P9US2=0 timeout 300 python tools/debug/p9us_time.py 2>&1 | grep p9us | sed 's/^/old /' >> $O/p9us_ab.log
  timeout 300 python tools/debug/p9us_time.py 2>&1 | grep p9us | sed 's/^/new /' >> $O/p9us_ab.log
done
[ -f $C/libjp_probe_T2.so ] && JP_LIB_PATH=$C/libjp_probe_T2.so timeout 300 python tools/debug/p9us_trace_steps.py > $O/p9us2_steps.log 2>&1
for rep in 1 2 3; do
  JP_P9US2=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > $O/step_old_$rep.json 2>> $O/step.err
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > $O/step_new_$rep.json 2>> $O/step.err
done
tail -3 $O/pytest_iconv.log; cat $O/p9us_ab.log; cat $O/p9us2_steps.log
for f in $O/step_*.json; do python -c "
import json
d=json.load(open('$f')); print('$f'.split('/')[-1], d['ms_per_step'], d['value'])"; done
