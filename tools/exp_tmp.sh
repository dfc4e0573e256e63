timeout -k 10 200 python -u tools/attn_sweep.py --variant=0,0x800 --rounds=5 --launches=100 256 12 197 64 256 16 197 64 2>&1 | grep -v amdgpu.ids
