#!/bin/bash
# Is a GEMM's per-CU speed limited by the chip (power, fabric) or by the CU?  Same per-CU work on 256 / 128 / 64 CUs.
set -u
run() { timeout -k 10 150 "$@" 2>&1 | grep -v amdgpu.ids; rc=${PIPESTATUS[0]}; case $rc in 124|134|137|139) echo "rc=$rc: stopping"; exit $rc;; esac; }
for epi in gelu resid; do
  if [ $epi = gelu ]; then N=3072; K=768; else N=768; K=3072; fi
  echo "== $epi 256 CUs"; run python -u tools/gemm_sweep.py --epi=$epi --rounds=3 --launches=200 50432 $N $K
  echo "== $epi 128 CUs, half the rows"; VITK_CU_LIMIT=128 run python -u tools/gemm_sweep.py --epi=$epi --rounds=3 --launches=200 25216 $N $K
  echo "== $epi 64 CUs, a quarter of the rows"; VITK_CU_LIMIT=64 run python -u tools/gemm_sweep.py --epi=$epi --rounds=3 --launches=200 12608 $N $K
done
echo "== 8192^3 256 CUs"; run python -u tools/gemm_sweep.py --rounds=3 --launches=30 8192 8192 8192
echo "== 4096x8192x8192 128 CUs"; VITK_CU_LIMIT=128 run python -u tools/gemm_sweep.py --rounds=3 --launches=30 4096 8192 8192
exit 0
