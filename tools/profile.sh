#!/bin/bash
# rocprofv3 passes on the GPU box: kernel trace + stats of bench.py, then PMC passes (each its own run)
# on the per-kernel microbenchmark.  Output under gpurun_out/prof_*.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
fatal() { case "$1" in 124|134|137|139) echo "step '$2' ended with $1: stopping"; exit "$1";; esac; }

timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o bench -- \
    python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/prof_bench.log 2>&1
rc=$?; tail -n 3 $OUT/prof_bench.log; echo "kernel-trace rc=$rc"; fatal $rc trace

i=0
for pmc in "$@"; do
    i=$((i+1))
    timeout -s KILL 180 rocprofv3 --pmc $pmc --output-format csv -d $OUT/prof_pmc$i -o pmc -- \
        python3 $ROOT/tools/microbench.py --iters 3 > $OUT/prof_pmc$i.log 2>&1
    rc=$?; echo "pmc pass $i ($pmc) rc=$rc"; tail -n 2 $OUT/prof_pmc$i.log; fatal $rc pmc$i
done
exit 0
