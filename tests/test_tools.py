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
