#!/bin/bash
# The whole GPU test suite + smoke + one bench line; stops at a fault, abort or time limit.
set -u
mkdir -p gpurun_out
fatal() { case "$1" in 124|134|137|139) echo "step '$2' ended with $1: stopping"; exit "$1";; esac; }
timeout -k 10 1100 python -u -m pytest tests -m gpu -q -x --timeout 300 --timeout-method thread -p no:cacheprovider "$@" > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -n 6 gpurun_out/pytest_gpu.log; echo "pytest rc=$rc"; fatal $rc pytest
[ $rc -ne 0 ] && exit $rc
timeout -k 10 120 python -u __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
rc=$?; tail -n 1 gpurun_out/smoke.log; fatal $rc smoke
timeout -k 10 400 python -u bench.py --steps 30 --warmup 8 > gpurun_out/bench_b16_224.log 2>&1; rc=$?; tail -n 1 gpurun_out/bench_b16_224.log; fatal $rc bench
exit $rc
