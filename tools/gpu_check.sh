#!/bin/bash
# One GPU-box session: parity tests, per-kernel timings, smoke, a short bench.  Stops at the first
# step that ends in a fault, abort or time limit; an ordinary test failure does not stop it.
#   tools/gpu_check.sh [microbench variant args ...]   e.g. "--gemm-variant 1" "--gemm-variant 0x43"
set -u
mkdir -p gpurun_out
fatal() { case "$1" in 124|134|137|139) echo "step '$2' ended with $1: stopping"; exit "$1";; esac; }

timeout -k 10 900 python -u -m pytest tests -m gpu -q --timeout 120 --timeout-method thread -p no:cacheprovider \
    > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -n 25 gpurun_out/pytest_gpu.log; echo "pytest rc=$rc"; fatal $rc pytest

timeout -k 10 120 python -u __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
rc=$?; tail -n 2 gpurun_out/smoke.log; echo "smoke rc=$rc"; fatal $rc smoke

i=0
for args in "" "$@"; do
    timeout -k 10 300 python -u tools/microbench.py $args > gpurun_out/microbench_$i.log 2>&1
    rc=$?; cat gpurun_out/microbench_$i.log | grep -v amdgpu.ids; fatal $rc "microbench $args"
    i=$((i+1))
done

timeout -k 10 400 python -u bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1
rc=$?; tail -n 2 gpurun_out/bench.log; echo "bench rc=$rc"; fatal $rc bench
exit 0
