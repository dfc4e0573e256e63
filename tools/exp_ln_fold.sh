#!/bin/bash
set -u
run() { timeout -k 10 300 "$@" 2>&1 | grep -v amdgpu.ids; rc=${PIPESTATUS[0]}; case $rc in 124|134|137|139) echo "rc=$rc: stopping"; exit $rc;; esac; }
run python -u -m pytest tests/test_gpu_kernels.py -q -x -s -k "folded" --timeout 120 --timeout-method thread -p no:cacheprovider | tail -12
run python -u -m pytest tests/test_gpu_model.py -q -x -s -k "pool or stages or logits" --timeout 300 --timeout-method thread -p no:cacheprovider | grep -E "pool|passed|failed|Error|error|bf16:" | tail -30
bash tools/exp_ab_env.sh VIT_AMD_NO_LN_FOLD=1 | grep -o '== [a-zA-Z_ =0-9]*\|"value": [0-9.]*\|"ms_per_step": [0-9.]*' | paste - - -
exit 0
