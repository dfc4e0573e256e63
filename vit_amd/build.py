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


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, probes="--probes" in sys.argv))
