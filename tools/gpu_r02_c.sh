#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_gpu_model.py tests/test_train_step.py tests/test_train_ddp.py tests/test_gpu_fp32.py -m gpu -q -s --timeout 200 --timeout-method thread -p no:cacheprovider > gpurun_out/pytest_pool.log 2>&1
rc=$?; grep -E "max-abs|passed|failed|Error|error" gpurun_out/pytest_pool.log | tail -40
timeout -k 10 120 python -u __graft_entry__.py smoke 2>&1 | tail -1
exit $rc
