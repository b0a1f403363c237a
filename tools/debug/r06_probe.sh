mkdir -p gpurun_out/r06q
timeout 400 python tests/probe_pose_branch.py --quick > gpurun_out/r06q/probe_fixed.log 2>&1
grep -h 'as-test' gpurun_out/r06q/probe_fixed.log | cut -c1-260
timeout 900 python -m pytest tests/test_batch_lifetime_gpu.py -q -m gpu -x > gpurun_out/r06q/lifetime.log 2>&1; tail -5 gpurun_out/r06q/lifetime.log
timeout 1200 python -m pytest tests/test_multi_step_parity_gpu.py -q -m gpu -s > gpurun_out/r06q/multi_step.log 2>&1; tail -5 gpurun_out/r06q/multi_step.log
timeout 600 python -m pytest tests/test_split_accuracy_gpu.py -q -m gpu -s -k one_scale > gpurun_out/r06q/outliers.log 2>&1; tail -3 gpurun_out/r06q/outliers.log
