#!/bin/bash
# rocprofv3 evidence for one workload on the GPU box:
#   tools/profile.sh NAME ARCH IMG BATCH        e.g.  tools/profile.sh b16_224_bs256 b16 224 256
# 1. kernel trace + stats of bench.py (the command whose line is reported), 2. three PMC passes, each its own
# run, on the per-kernel microbenchmark at the same shapes (SQ counters; FETCH_SIZE + GRBM_GUI_ACTIVE; WRITE_SIZE).
# Output under gpurun_out/prof_NAME/; tools/pmc_summary.py condenses it (and writes the traffic table bench.py reads).
set -u
NAME=$1; ARCH=$2; IMG=$3; BATCH=$4
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
fatal() { case "$1" in 124|134|137|139) echo "step '$2' ended with $1: stopping"; exit "$1";; esac; }

timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- \
    python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --arch $ARCH --img-size $IMG --batch $BATCH > $OUT/bench_under_rocprof.log 2>&1
rc=$?; grep "^{" $OUT/bench_under_rocprof.log | tail -n 1 > $OUT/bench_line_under_rocprof.json; echo "kernel-trace rc=$rc"; fatal $rc trace

i=0
for pmc in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout -s KILL 180 rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc$i -o pmc -- \
        python3 $ROOT/tools/microbench.py --iters 3 --arch $ARCH --img-size $IMG --batch $BATCH > $OUT/pmc$i.log 2>&1
    rc=$?; echo "pmc pass $i rc=$rc"; fatal $rc pmc$i
done
cd $ROOT
python3 tools/pmc_summary.py --workload $NAME $OUT/trace $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 > $OUT/summary.txt
head -n 12 $OUT/summary.txt
exit 0
