"""The signed-digit extraction of the multiexp (msm_impl.hpp: power-of-two and mixed-radix windows) on the HOST, through the
library's self-test hook: (1) the digits reconstruct the scalar, sum_w d_w * weight_w == k, with every digit in its signed
range; (2) a window-group rank's digits -- chain started one window below its first window, fallback on the sign boundary --
equal the plain chain's for every group start, on random scalars and on scalars crafted to sit ON the boundary (raw digit of
the window below == nb, with and without a carry coming from further down)."""
import ctypes as C
import random

import numpy as np
import pytest

import bn254_model as M

R = M.R_ORDER
INT_MIN = -(1 << 31)


@pytest.fixture(scope="module")
def lib():
    import phase2_bn254_amd as zk

    return zk.lib.load()


def _geom(lib, n, wgroups):
    g = np.zeros(5 + 128, np.uint32)
    assert lib.mi355zk_selftest_msm_digits(n, wgroups, None, 0, 0, 0, None, g.ctypes.data) == 0
    c, W, nb, rmul, rshift = (int(x) for x in g[:5])
    return {"c": c, "W": W, "nb": nb, "rmul": rmul, "rshift": rshift, "width": [int(x) for x in g[5:5 + W]], "shift": [int(x) for x in g[5 + W:5 + 2 * W]]}


def _weights(G):
    if G["rmul"] != 1:
        B = G["rmul"] << G["rshift"]
        return [B ** w for w in range(G["W"])], [B] * G["W"]
    return [1 << s for s in G["shift"]], [1 << wd for wd in G["width"]]


def _digits(lib, n, wgroups, k, w_start, w_stop, direct, W):
    sc = np.array([(k >> (32 * i)) & 0xFFFFFFFF for i in range(8)], dtype=np.uint32)
    out = np.zeros(W, np.int32)
    assert lib.mi355zk_selftest_msm_digits(n, wgroups, sc.ctypes.data, w_start, w_stop, direct, out.ctypes.data, None) == 0
    return [int(x) for x in out]


# sizes that select different layouts: power-of-two windows (small n) and mixed-radix windows of several multipliers
SIZES = [(1 << 10, 1), (1 << 16, 1), (1 << 20, 1), (1 << 22, 2), (1 << 24, 4), (1 << 26, 4), (1 << 26, 2), (1 << 27, 4), (1 << 28, 1),
         (20480, 1), (98304, 1), (40960, 2), (14336, 1), (12582912, 1), (256, 4), (8388608, 1)]   # multipliers 3, 7, 9, 11, 13 (13 and 24 windows), 5


@pytest.mark.parametrize("n,wgroups", SIZES)
def test_digits_reconstruct_the_scalar_and_group_starts_agree(lib, n, wgroups):
    G = _geom(lib, n, wgroups)
    W, nb = G["W"], G["nb"]
    weight, base = _weights(G)
    rnd = random.Random(n * 7 + wgroups)
    scalars = [0, 1, 2, R - 1, R - 2, (1 << 253) + 12345, (1 << 128) - 1] + [rnd.randrange(R) for _ in range(60)]
    # boundary scalars: raw digit of window v exactly on the sign boundary (nb for mixed radix, 2^(width-1) for power-of-two
    # windows), everything above random, the window below either far from a carry (0), just producing one (boundary + 1) or itself
    # on the boundary behind a carry (a chain two windows long)
    for v in range(0, W - 1):
        half = nb if G["rmul"] != 1 else base[v] // 2
        for below in ("none", "carry", "chain"):
            k = half * weight[v]
            if v >= 1 and below != "none":
                hb = nb if G["rmul"] != 1 else base[v - 1] // 2
                k += (hb + 1 if below == "carry" else hb) * weight[v - 1]
                if below == "chain" and v >= 2:
                    hb2 = nb if G["rmul"] != 1 else base[v - 2] // 2
                    k += (hb2 + 1) * weight[v - 2]
            if v + 1 < W:
                top = (R - 1 - k) // weight[v + 1]
                if top > 0:
                    k += rnd.randrange(top) * weight[v + 1]
            if k < R:
                scalars.append(k)
    for k in scalars:
        full = _digits(lib, n, wgroups, k, 0, W, 0, W)
        assert INT_MIN not in full
        assert sum(d * wt for d, wt in zip(full, weight)) == k, (n, k)
        for w in range(W - 1):
            assert -(base[w] // 2) <= full[w] <= base[w] // 2 and abs(full[w]) <= nb
        assert 0 <= full[W - 1] <= nb
        for w_start in range(1, W):
            for w_stop in {W, min(W, w_start + 3)}:
                got = _digits(lib, n, wgroups, k, w_start, w_stop, 1, W)
                want = [d if w_start <= w < w_stop else INT_MIN for w, d in enumerate(full)]
                assert got == want, (n, wgroups, hex(k), w_start, w_stop)
