#!/bin/bash
# training path session: tests of the step, then the step's bench and its kernel trace
set -u
mkdir -p gpurun_out
fatal() { case "$1" in 124|134|137|139) echo "step '$2' ended with $1: stopping"; exit "$1";; esac; }
timeout -k 10 600 python -u -m pytest tests/test_train_step.py tests/test_abi.py -m "gpu or not gpu" -q -x --timeout 120 --timeout-method thread -p no:cacheprovider \
    > gpurun_out/pytest_train.log 2>&1
rc=$?; tail -n 30 gpurun_out/pytest_train.log; echo "pytest rc=$rc"; fatal $rc pytest
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u bench.py --steps 10 --warmup 3 --mode train --no-cpu-baseline > gpurun_out/bench_train.log 2>&1
rc=$?; tail -n 1 gpurun_out/bench_train.log; fatal $rc train
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_train -o train -- \
    python3 $ROOT/bench.py --steps 5 --warmup 2 --mode train --no-cpu-baseline > $ROOT/gpurun_out/prof_train.log 2>&1
rc=$?; tail -n 1 $ROOT/gpurun_out/prof_train.log; fatal $rc proftrain
cd $ROOT; python3 tools/pmc_summary.py gpurun_out/prof_train | head -40
exit 0
