#!/bin/bash
# The bench lines of the measured configurations on one box (no tests, no profiler).
set -u
mkdir -p gpurun_out
fatal() { case "$1" in 124|134|137|139) echo "step '$2' ended with $1: stopping"; exit "$1";; esac; }
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d.get('cls_only_last_block') or {}
print('$1', 'images/s', d['value'], 'ms/step', d['ms_per_step'], 'mfma_frac', d.get('mfma_peak_frac'), 'roofline', d.get('roofline',{}).get('avg_launch_ms'), d.get('roofline',{}).get('frac'), 'cls-only', c.get('value'))"; }
timeout -k 10 400 python -u bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | line b16_224; fatal ${PIPESTATUS[0]} b16
timeout -k 10 300 python -u bench.py --steps 10 --warmup 3 --arch l16 --no-cpu-baseline 2>/dev/null | line l16_224; fatal ${PIPESTATUS[0]} l16
timeout -k 10 300 python -u bench.py --steps 10 --warmup 3 --img-size 384 --batch 128 --no-cpu-baseline 2>/dev/null | line b16_384; fatal ${PIPESTATUS[0]} b384
timeout -k 10 300 python -u bench.py --steps 10 --warmup 3 --arch s16 --no-cpu-baseline 2>/dev/null | line s16_224; fatal ${PIPESTATUS[0]} s16
timeout -k 10 300 python -u bench.py --steps 10 --warmup 3 --mode train --no-cpu-baseline 2>/dev/null | line train_b16; fatal ${PIPESTATUS[0]} train
exit 0
