"""The measurement scripts under tools/ are not run by the suite; this keeps them parseable."""
import glob
import os
import py_compile
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_python_tools_compile(tmp_path):
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")))
    assert files
    for f in files:
        py_compile.compile(f, cfile=str(tmp_path / (os.path.basename(f) + "c")), doraise=True)


def test_shell_tools_parse():
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh")))
    assert files
    for f in files:
        subprocess.run(["bash", "-n", f], check=True)


def test_counted_wait_scanner_flags_a_scalar_load_inside_the_region():
    """The attention and weight-gradient kernels wait for inline-asm LDS reads with counted lgkmcnt waits; the build
    fails if the compiler puts a scalar-memory instruction (which returns out of order on the same counter) between a
    transposed LDS read and the wait that drains the counter."""
    from vit_amd import build
    clean = """
_Z6kernelPv:
\tds_read_b64_tr_b16 v[0:1], v2 offset:0
\tds_read_b64_tr_b16 v[4:5], v2 offset:1024
\ts_waitcnt lgkmcnt(1)
\tv_mfma_f32_32x32x16_bf16 v[8:23], v[0:3], v[4:7], v[8:23]
\ts_waitcnt lgkmcnt(0)
\ts_load_dwordx2 s[0:1], s[4:5], 0x0
"""
    assert build.scan_counted_waits(clean) == []
    dirty = clean.replace("\ts_waitcnt lgkmcnt(1)\n", "\ts_load_dword s7, s[4:5], 0x10\n\ts_waitcnt lgkmcnt(1)\n")
    found = build.scan_counted_waits(dirty)
    assert len(found) == 1 and found[0][0] == "_Z6kernelPv" and "s_load_dword" in found[0][2]
    # a new kernel starts clean even if the previous one ended inside a region
    two = "_Z1aPv:\n\tds_read_b64_tr_b16 v[0:1], v2\n_Z1bPv:\n\ts_load_dword s0, s[4:5], 0x0\n\ts_waitcnt lgkmcnt(0)\n"
    assert build.scan_counted_waits(two) == []


def test_shipped_kernels_have_no_scalar_load_inside_counted_waits():
    from vit_amd import build
    assert build.check_counted_waits(verbose=False)


def test_two_plane_representation_of_f32_round_trips_every_bit_pattern_class():
    """The residual stream between the folded blocks is held as plane H = upper half of bits(x) + 0x8000 (bf16 rounded
    half up) and plane L = lower half of bits(x); bits(x) = (H << 16) + sext16(L).  Host-side helpers (torch ops, the
    GPU epilogues write the same planes): exact on ties, carries, zeros, denormals, and plane H is the nearest bf16
    except that ties go up."""
    import torch
    from vit_amd import ops
    g = torch.Generator().manual_seed(3)
    bits = torch.randint(-2**31, 2**31 - 1, (200000,), generator=g, dtype=torch.int64).to(torch.int32)
    exp = (bits >> 23) & 0xFF
    bits = bits[exp != 0xFF]                                         # finite values
    special = torch.tensor([0, -2**31, 0x3F800000, 0x3F808000, 0x3F818000, 0x3F80FFFF, 0x3FFFFFFF, 0x00000001, 0x007FFFFF,
                            0x3F807FFF, 0x3F808001], dtype=torch.int64).to(torch.int32)
    x = torch.cat([bits, special]).view(torch.float32)
    h, lo = ops.f32_to_planes(x)
    assert h.dtype == torch.bfloat16 and lo.dtype == torch.bfloat16
    back = ops.planes_to_f32(h, lo)
    assert torch.equal(back.view(torch.int32), x.view(torch.int32))
    near = x.to(torch.bfloat16)                                      # round to nearest even
    tie = (x.view(torch.int32) & 0xFFFF) == 0x8000
    same = h.view(torch.int16) == near.view(torch.int16)
    big = ((x.view(torch.int32) >> 16) & 0x7FFF) >= 0x7F7F           # rounds to infinity either way: leave out
    assert bool((same | tie | big).all())
    up = (h.float() - x).abs() <= (near.float() - x).abs() * (1 + 1e-6) + 0.0
    assert bool((up | big | ~torch.isfinite(h.float())).all())       # never farther than the nearest bf16
