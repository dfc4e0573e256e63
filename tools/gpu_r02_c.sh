#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_gpu_model.py tests/test_train_step.py tests/test_gpu_fp32.py -m gpu -q -x --timeout 200 --timeout-method thread -p no:cacheprovider > gpurun_out/pytest_pool.log 2>&1
rc=$?; tail -n 5 gpurun_out/pytest_pool.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u bench.py --steps 30 --warmup 8 --no-cpu-baseline | tail -1
cd /tmp; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1; grep -E "gemm_f32|small" /tmp/pt/*/b_kernel_stats.csv /tmp/pt/b_kernel_stats.csv 2>/dev/null | cut -c1-200
