#!/bin/bash
# Same-box A/B of an environment switch on the whole forward: tools/exp_ab_env.sh VAR=VALUE [bench args...]
# runs bench.py alternately with and without the variable, three times each.
set -u
SW=$1; shift
run() { timeout -k 10 200 "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-175; rc=${PIPESTATUS[0]}; case $rc in 124|134|137|139) echo "rc=$rc: stopping"; exit $rc;; esac; }
for i in 1 2 3; do
  echo "== with $SW";    run env $SW python -u bench.py --steps 30 --warmup 8 --no-cpu-baseline "$@"
  echo "== without";     run python -u bench.py --steps 30 --warmup 8 --no-cpu-baseline "$@"
done
exit 0
