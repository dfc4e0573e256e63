#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_train_step.py -m gpu -q -x --timeout 200 --timeout-method thread -p no:cacheprovider > gpurun_out/pytest_f.log 2>&1
rc=$?; tail -n 4 gpurun_out/pytest_f.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python -u tools/tn_sweep.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python -u bench.py --steps 10 --warmup 3 --mode train --no-cpu-baseline | tail -1 | cut -c1-220
