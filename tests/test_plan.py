"""The tiled GEMM's tile planner (host logic of the library, vit_amd/csrc/gemm_plan.h) through vitk_gemm_plan:
runs without a GPU.  Invariants over many shapes, and the choices the forced-height sweeps on the MI355X found
best for the ViT shapes of the three forward configurations."""
import ctypes
import random

import pytest

from vit_amd import lib

KEYS = ("mtot", "tiles_m", "tiles_n", "gn", "mixed", "xcd_units", "ha", "na", "hb", "nb")


def plan(M, N, K, cus=256):
    L = lib.load()
    out = (ctypes.c_int * 10)()
    rc = L.vitk_gemm_plan(M, N, K, cus, out)
    assert rc == 0, L.vitk_last_error()
    return dict(zip(KEYS, out))


def busiest_workgroup(p, slots=32):
    """m-tiles of the busiest workgroup of an XCD under the two-height plan (workgroup s takes tiles s, s + 32, ...)."""
    tiles = (p["na"] + p["nb"]) * p["tiles_n"]
    loads = [sum(p["ha"] if (t // p["tiles_n"]) < p["na"] else p["hb"] for t in range(s, tiles, slots)) for s in range(slots)]
    return max(loads)


def _shapes(n, seed):
    rng = random.Random(seed)
    return [(rng.randint(128, 120000), 256 * rng.randint(1, 16), 64 * rng.choice([2, 4, 12, 16, 48, 64, 3, 5])) for _ in range(n)]


@pytest.mark.parametrize("M,N,K", _shapes(60, 7))
def test_plan_invariants(M, N, K):
    p = plan(M, N, K)
    assert 10 <= p["mtot"] <= 16
    assert p["tiles_n"] == N // 256 and p["tiles_m"] == -(-M // (16 * p["mtot"]))
    assert 0 <= p["gn"] < max(p["tiles_n"], 1)
    if not p["mixed"]:
        assert p["ha"] == p["na"] == p["hb"] == p["nb"] == p["xcd_units"] == 0
        return
    # two-height plan: plain tile order (few n-tiles, or weight panels so small that all stay in L2), the two-group
    # kernel's shapes, at least one tile per CU
    assert p["gn"] == 0 and (K // 64) % 2 == 0 and p["tiles_m"] * p["tiles_n"] >= 256
    assert 16 >= p["ha"] >= p["hb"] >= 10 and p["na"] >= 0 and p["nb"] >= 0
    units = p["xcd_units"]
    assert units * 8 * 16 >= M > (units - 1) * 8 * 16 - 8 * 16           # the 8 XCDs' row ranges cover M
    covered = p["na"] * p["ha"] + p["nb"] * p["hb"]
    assert units <= covered < units + max(p["ha"], p["hb"])               # panels cover the XCD's rows, none spare
    # and no workgroup is busier than under equal tiles of the planner's own uniform choice
    uniform = -(-(p["tiles_m"] * p["tiles_n"]) // 256) * p["mtot"]
    assert busiest_workgroup(p) <= uniform


def test_plan_choices_for_the_vit_shapes():
    # B/16 224 (M = 256 x 197): out_proj / fc2 take the two-height plan, 37 m-tiles on the busiest workgroup instead of 39
    for K in (768, 3072):
        p = plan(50432, 768, K)
        assert p["mixed"] == 1 and busiest_workgroup(p) == 37
    assert plan(50432, 3072, 768)["mtot"] == 15 and plan(50432, 3072, 768)["mixed"] == 0      # fc1: 10 rounds of 15
    assert plan(50432, 2304, 768)["mtot"] == 16                                                # qkv: 7 rounds of 16
    # B/16 384 (M = 128 x 577): fc1 at 16 m-tiles x 14 rounds (328 us), not 11 x 20 (361 us)
    assert plan(73856, 3072, 768)["mtot"] == 16
    # L/16 (D = 1024): fc1 at 16 m-tiles (376 us; 408 us at 11)
    assert plan(50432, 4096, 1024)["mtot"] == 16
    # small problems: more, shorter tiles rather than a few tall ones
    assert plan(1576, 768, 768)["mtot"] == 10 and plan(1576, 768, 768)["mixed"] == 0
    # a K the two-group kernel does not take (K / 64 odd): equal tiles
    assert plan(50432, 768, 192)["mixed"] == 0


def test_plan_rejects_bad_shapes():
    L = lib.load()
    out = (ctypes.c_int * 10)()
    assert L.vitk_gemm_plan(100, 100, 64, 0, out) == 2
    assert L.vitk_gemm_plan(100, 256, 64, 0, None) == 1
