#!/bin/bash
# Kernel trace + stats of bench.py only (no PMC passes): tools/trace_only.sh NAME [bench args...]
set -u
NAME=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- \
    python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $OUT/bench.log 2>&1
rc=$?; echo "trace rc=$rc"
cut -d, -f1-8 $OUT/bench_kernel_stats.csv | head -n 24
exit $rc
