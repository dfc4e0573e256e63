#!/bin/bash
# GPU-box session: the rocprofv3 evidence (kernel trace + PMC passes) of the three measured forward configurations and the
# kernel trace of the training step; tools/collect_profiles.py then copies the summaries into profiles/.
set -u
bash tools/profile.sh b16_224_bs256 b16 224 256 || exit $?
bash tools/profile.sh l16_224_bs256 l16 224 256 || exit $?
bash tools/profile.sh b16_384_bs128 b16 384 128 || exit $?
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_train_b16_224_bs256
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- \
    python3 $ROOT/bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
rc=$?; grep "^{" $OUT/bench_under_rocprof.log | tail -n 1 > $OUT/bench_line_under_rocprof.json; echo "train kernel-trace rc=$rc"
exit $rc
