#!/bin/bash
# round-5 GPU session A: X variants (4-wave workgroups on the 256-row banks), s_setprio build, timing probes, P9US step trace
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=$ROOT/gpurun_out/r05a; mkdir -p $O
C=$ROOT/jperceiver_amd/csrc
CASES="merge 256->256 3x3 refl @256,merge 256->256 3x3 refl @128,CRP 256->256 1x1 @256,layer3 256->256 3x3 @64"
JP_P9_X=1 JP_P1_X=1 timeout 300 python tools/debug/p9_check.py > $O/x_check.log 2>&1
JP_P9_X=1 JP_P1_X=1 timeout 300 python tools/debug/p1_check.py >> $O/x_check.log 2>&1
timeout 300 python tools/debug/p9_check.py >> $O/x_check.log 2>&1
for rep in 1 2; do
  for lib in default PRIO AHALF NOB NOSTG; do
    L="JP_NONE=1"; [ $lib != default ] && L="JP_LIB_PATH=$C/libjp_probe_$lib.so"
    env $L timeout 300 python tools/debug/p9us_time.py >> $O/p9us_$lib.log 2>&1
    env $L timeout 300 python tools/conv_bench.py --iters 3 --only "$CASES" >> $O/conv_$lib.log 2>&1
  done
  JP_P9_X=1 JP_P1_X=1 timeout 300 python tools/conv_bench.py --iters 3 --only "$CASES" >> $O/conv_X.log 2>&1
done
JP_LIB_PATH=$C/libjp_probe_TS.so timeout 300 python tools/debug/p9us_trace_steps.py > $O/p9us_steps.log 2>&1
# whole step, same box: default vs X
for rep in 1 2; do
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > $O/step_default_$rep.json 2>> $O/step.err
  JP_P9_X=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > $O/step_X9_$rep.json 2>> $O/step.err
  JP_P9_X=1 JP_P1_X=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > $O/step_X91_$rep.json 2>> $O/step.err
done
tail -4 $O/x_check.log; grep -h "p9us" $O/p9us_*.log | tail -20; cat $O/p9us_steps.log; for f in $O/step_*.json; do echo $f; cut -c1-160 $f; done
