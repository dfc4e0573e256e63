#!/bin/bash
# Round-end session on the GPU box: the whole GPU test suite, smoke, the bench lines of the measured
# configurations, and the rocprofv3 evidence for each (tools/profile.sh).
set -u
mkdir -p gpurun_out
fatal() { case "$1" in 124|134|137|139) echo "step '$2' ended with $1: stopping"; exit "$1";; esac; }
timeout -k 10 1000 python -u -m pytest tests -m gpu -q -x --timeout 300 --timeout-method thread -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -n 4 gpurun_out/pytest_gpu.log; echo "pytest rc=$rc"; fatal $rc pytest
[ $rc -ne 0 ] && exit $rc
timeout -k 10 120 python -u __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
rc=$?; tail -n 1 gpurun_out/smoke.log; fatal $rc smoke
timeout -k 10 400 python -u bench.py --steps 30 --warmup 8 > gpurun_out/bench_b16_224.log 2>&1; rc=$?; tail -n 1 gpurun_out/bench_b16_224.log; fatal $rc bench
timeout -k 10 300 python -u bench.py --steps 10 --warmup 3 --arch l16 --no-cpu-baseline > gpurun_out/bench_l16.log 2>&1; rc=$?; tail -n 1 gpurun_out/bench_l16.log | cut -c1-260; fatal $rc l16
timeout -k 10 300 python -u bench.py --steps 10 --warmup 3 --img-size 384 --batch 128 --no-cpu-baseline > gpurun_out/bench_384.log 2>&1; rc=$?; tail -n 1 gpurun_out/bench_384.log | cut -c1-260; fatal $rc b384
timeout -k 10 300 python -u bench.py --steps 10 --warmup 3 --arch s16 --no-cpu-baseline > gpurun_out/bench_s16.log 2>&1; rc=$?; tail -n 1 gpurun_out/bench_s16.log | cut -c1-260; fatal $rc s16
timeout -k 10 300 python -u bench.py --steps 10 --warmup 3 --mode train --no-cpu-baseline > gpurun_out/bench_train.log 2>&1; rc=$?; tail -n 1 gpurun_out/bench_train.log | cut -c1-260; fatal $rc train
VIT_AMD_SHARE_GPU=1 timeout -k 10 300 python -u bench.py --gpus 2 --steps 5 --warmup 2 --batch 64 --no-cpu-baseline > gpurun_out/bench_2rank_shared.log 2>&1; rc=$?; tail -n 1 gpurun_out/bench_2rank_shared.log | cut -c1-260; fatal $rc gpus2
exit 0
