python -u tools/gemm_sweep.py --epi=resid --rounds=3 --launches=100 50432 768 3072 50432 768 768 2>&1 | grep variant
python -u tools/gemm_sweep.py --epi=residln --rounds=3 --launches=100 50432 768 3072 50432 768 768 2>&1 | grep variant
