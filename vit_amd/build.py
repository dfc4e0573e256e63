"""Build libvitk.so (the HIP kernels + C ABI) in-tree with plain hipcc.

    python -m vit_amd.build [--force]

No torch headers are involved: the library is a bare `hipcc -shared -fPIC
--offload-arch=gfx950` product, loaded with ctypes (vit_amd/lib.py).  hipcc
cross-compiles gfx950 code without a GPU.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvitk.so")
SOURCES = ["vitk_api.hip", "gemm.hip", "gemm_pipe.hip", "gemm_pp.hip", "attention.hip", "rowops.hip", "f32path.hip", "backward.hip", "gemm_tn.hip", "attention_bwd.hip"]
HEADERS = [os.path.join(CSRC, "vitk_common.h"), os.path.join(CSRC, "vitk_kernels.h"), os.path.join(CSRC, "gemm_common.h"), os.path.join(CSRC, "gemm_plan.h"),
           os.path.join(os.path.dirname(HERE), "include", "vitk.h")]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=fast"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, probes=False):
    """Compile every .hip source to an object (in parallel) and link libvitk.so.  probes=True builds the
    diagnostic library libvitk_probes.so instead (-DVITK_PROBES: timing probes that give wrong or no results,
    for tools/ only; select it with VITK_LIB=...)."""
    objdir = os.path.join(HERE, "build_probes" if probes else "build")
    lib = os.path.join(HERE, "libvitk_probes.so") if probes else LIB
    flags = FLAGS + (["-DVITK_PROBES"] if probes else [])
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + HEADERS):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + flags + ["-c", s, "-o", o]
        if verbose:
            print("[vit_amd.build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(lib, objs):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib] + objs
        if verbose:
            print("[vit_amd.build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return lib


# Sources whose kernels wait for inline-asm LDS reads with counted `s_waitcnt lgkmcnt(n)`: the counts hold only while
# nothing else of the wave is outstanding on that counter.  LDS returns in order, scalar memory does not, so one scalar
# load (a kernel-argument or descriptor reload) between a transposed LDS read and the wait that covers it could let
# the wait pass early.
COUNTED_WAIT_SOURCES = ["attention.hip", "gemm_tn.hip"]


def scan_counted_waits(asm_text):
    """(kernel, line number, instruction) of every scalar-memory instruction that sits, in program text order, between
    a ds_read_b64_tr_b16 and the next `s_waitcnt` with lgkmcnt(0) of the same kernel."""
    import re
    found, kernel, pending = [], None, False
    for ln, line in enumerate(asm_text.splitlines(), 1):
        t = line.strip()
        if re.match(r"^_Z\w+:", t):
            kernel, pending = t[:-1], False
            continue
        if not t or t[0] in ";.":
            continue
        ins = t.split()[0]
        if ins == "ds_read_b64_tr_b16":
            pending = True
        elif ins == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m and int(m.group(1)) == 0:
                pending = False
        elif pending and ins.startswith(("s_load_", "s_buffer_load_", "s_memtime", "s_memrealtime", "s_dcache")):
            found.append((kernel, ln, t))
    return found


def check_counted_waits(verbose=True):
    """Compile COUNTED_WAIT_SOURCES to gfx950 assembly (cached beside the objects) and fail if the compiler put a
    scalar-memory instruction inside a counted-wait region."""
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()

    def asm_of(src):
        s, out = os.path.join(CSRC, src), os.path.join(objdir, src.replace(".hip", ".s"))
        if _stale(out, [s] + HEADERS):
            cmd = [hipcc] + FLAGS + ["-S", "--cuda-device-only", s, "-o", out]
            if verbose:
                print("[vit_amd.build]", " ".join(cmd), flush=True)
            subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        return out

    with ThreadPoolExecutor(max_workers=len(COUNTED_WAIT_SOURCES)) as ex:
        outs = list(ex.map(asm_of, COUNTED_WAIT_SOURCES))
    for src, out in zip(COUNTED_WAIT_SOURCES, outs):
        with open(out) as f:
            bad = scan_counted_waits(f.read())
        if bad:
            k, ln, ins = bad[0]
            raise RuntimeError(f"{src}: {len(bad)} scalar-memory instruction(s) inside a counted lgkmcnt wait region, "
                               f"first in {k} at {os.path.basename(out)}:{ln}: {ins}")
    return True


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, probes="--probes" in sys.argv))
    if "--check-waits" in sys.argv:
        print("counted waits clean:", check_counted_waits())
