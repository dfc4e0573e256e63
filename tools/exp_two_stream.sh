#!/bin/bash
# GPU session of the micro-batch stream experiment: baseline bench line, then S streams x CU limits.
set -u
mkdir -p gpurun_out
run() { timeout -k 10 150 "$@" 2>&1 | grep -v amdgpu.ids; rc=${PIPESTATUS[0]}; case $rc in 124|134|137|139) echo "rc=$rc: stopping"; exit $rc;; esac; }
run python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline | cut -c1-200
run python -u tools/two_stream.py --streams 1
VITK_CU_LIMIT=128 run python -u tools/two_stream.py --streams 2
VITK_CU_LIMIT=128 run python -u tools/two_stream.py --streams 2 --delay-us 150
VITK_CU_LIMIT=128 run python -u tools/two_stream.py --streams 2 --delay-us 400
run python -u tools/two_stream.py --streams 2
VITK_CU_LIMIT=192 run python -u tools/two_stream.py --streams 2 --delay-us 150
VITK_CU_LIMIT=64 run python -u tools/two_stream.py --streams 4 --delay-us 100
VITK_CU_LIMIT=128 run python -u tools/two_stream.py --streams 4 --delay-us 100
run python -u tools/two_stream.py --streams 1
exit 0
