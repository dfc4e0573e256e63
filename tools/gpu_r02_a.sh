#!/bin/bash
# GPU-box session: parity tests, the no-alloc entry point's test, smoke, then forward throughput against the
# per-GPU batch size (is a smaller working set faster per image?) and the other measured configurations.
set -u
mkdir -p gpurun_out
fatal() { case "$1" in 124|134|137|139) echo "step '$2' ended with $1: stopping"; exit "$1";; esac; }

timeout -k 10 900 python -u -m pytest tests -m gpu -q -x --timeout 120 --timeout-method thread -p no:cacheprovider \
    > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -n 8 gpurun_out/pytest_gpu.log; echo "pytest rc=$rc"; fatal $rc pytest

timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -m gpu -q -x --timeout 120 --timeout-method thread -p no:cacheprovider \
    > gpurun_out/pytest_noalloc.log 2>&1
rc=$?; tail -n 8 gpurun_out/pytest_noalloc.log; echo "no-alloc rc=$rc"; fatal $rc noalloc

timeout -k 10 120 python -u __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
rc=$?; tail -n 2 gpurun_out/smoke.log; echo "smoke rc=$rc"; fatal $rc smoke

for b in 256 128 64 512 256; do
    timeout -k 10 300 python -u bench.py --steps 30 --warmup 8 --batch $b --no-cpu-baseline > gpurun_out/bench_b$b.log 2>&1
    rc=$?; tail -n 1 gpurun_out/bench_b$b.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch', d['config']['per_gpu_batch'], 'img/s', d['value'], 'ms', d['ms_per_step'], 'fc1 frac', d['roofline']['frac'])"
    fatal $rc "bench $b"
done
timeout -k 10 300 python -u bench.py --steps 10 --warmup 3 --arch l16 --no-cpu-baseline > gpurun_out/bench_l16.log 2>&1
rc=$?; tail -n 1 gpurun_out/bench_l16.log; fatal $rc l16
timeout -k 10 300 python -u bench.py --steps 10 --warmup 3 --img-size 384 --batch 128 --no-cpu-baseline > gpurun_out/bench_384.log 2>&1
rc=$?; tail -n 1 gpurun_out/bench_384.log; fatal $rc b384
timeout -k 10 300 python -u bench.py --steps 10 --warmup 3 --mode train --no-cpu-baseline > gpurun_out/bench_train.log 2>&1
rc=$?; tail -n 1 gpurun_out/bench_train.log; fatal $rc train
exit 0
