"""Pin oracle/bsmm_oracle.py to fixtures generated from the reference itself (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

from oracle import bsmm_oracle as orc

LUT_CASES = ["rand128", "ba160", "ba160_bs8", "holes", "single", "rect"]
MATH_LAYOUTS = ["ba16", "holes", "rect", "single"]


@pytest.fixture(scope="module")
def luts(golden_dir):
    return np.load(os.path.join(golden_dir, "luts.npz"))


@pytest.fixture(scope="module")
def math(golden_dir):
    return np.load(os.path.join(golden_dir, "math.npz"))


@pytest.mark.parametrize("name", LUT_CASES)
@pytest.mark.parametrize("z", [1, 0])
def test_oracle_luts_match_reference(luts, name, z):
    g = lambda k: luts["%s/z%d/%s" % (name, z, k)]
    t = orc.build_layout_luts(g("layout"), int(g("bsize")), bool(z))
    assert t["blocks"] == int(g("blocks"))
    for key in ("fprop_lut", "bprop_lut", "updat_lut", "l2_lut"):
        np.testing.assert_array_equal(t[key], g(key), err_msg=key)
    for key in ("fprop_segments", "bprop_segments", "fprop_locks", "bprop_locks", "fprop_shared",
                "bprop_shared", "l2_shared"):
        assert t[key] == int(g(key)), key


def test_known_numbers_from_survey(luts):
    # random(128,128) seed 0 and BA(160,5,seed=1)+I
    g = lambda n, k: luts["%s/z1/%s" % (n, k)]
    assert int(g("rand128", "blocks")) == 8292 and int(g("rand128", "fprop_segments")) == 128
    assert int(g("rand128", "fprop_locks")) == 0 and int(g("rand128", "fprop_shared")) == 640
    assert g("rand128", "fprop_lut")[:4].tolist() == [256, 51, 0, 0]
    assert int(g("ba160", "blocks")) == 1722 and int(g("ba160", "fprop_segments")) == 190
    assert int(g("ba160", "fprop_locks")) == 21 and int(g("ba160", "fprop_shared")) == 128


@pytest.mark.parametrize("lay", MATH_LAYOUTS)
@pytest.mark.parametrize("bs", [8, 16, 32])
@pytest.mark.parametrize("axis", [0, 1])
def test_oracle_math_matches_reference(math, lay, bs, axis):
    g = lambda k: math["%s/bs%d/a%d/%s" % (lay, bs, axis, k)]
    t = orc.build_layout_luts(g("layout"), bs)
    W, X, E = (g(k).astype(np.float32) for k in ("W", "X", "E"))
    for name, got, want in (("Y", orc.fprop(t, X, W, axis), g("Y")),
                            ("DX", orc.bprop(t, E, W, axis), g("DX")),
                            ("DW", orc.updat(t, X, E, axis), g("DW"))):
        want = want.astype(np.float64)
        scale = np.abs(want).max() + 1e-30
        assert np.abs(got - want).max() / scale < 1e-6, name     # fixture is float32-rounded
        # fast (batched BLAS) variants agree with the loop oracle
    np.testing.assert_allclose(orc.fprop_fast(t, X, W, axis, np.float64), orc.fprop(t, X, W, axis), atol=1e-12)
    np.testing.assert_allclose(orc.bprop_fast(t, E, W, axis, np.float64), orc.bprop(t, E, W, axis), atol=1e-12)
    np.testing.assert_allclose(orc.updat_fast(t, X, E, axis, np.float64), orc.updat(t, X, E, axis), atol=1e-12)


def test_oracle_equals_dense(math):
    g = lambda k: math["ba16/bs16/a1/%s" % k]
    t = orc.build_layout_luts(g("layout"), 16)
    W, X, E = (g(k).astype(np.float64) for k in ("W", "X", "E"))
    Wd = orc.to_dense(t, W)
    np.testing.assert_allclose(orc.fprop(t, X, W, 1), X @ Wd, atol=1e-12)
    np.testing.assert_allclose(orc.bprop(t, E, W, 1), E @ Wd.T, atol=1e-12)
    np.testing.assert_allclose(orc.fprop(t, X.T.copy(), W, 0), Wd.T @ X.T, atol=1e-12)
    np.testing.assert_allclose(orc.bprop(t, E.T.copy(), W, 0), Wd @ E.T, atol=1e-12)
    # updat = block gather of the dense outer product; alpha/beta/multi-pair semantics
    full = X.T @ E
    U = orc.updat(t, X, E, 1)
    for w, (c, k) in enumerate(t["updat_list"]):
        np.testing.assert_allclose(U[w], full[c * 16:(c + 1) * 16, k * 16:(k + 1) * 16], atol=1e-12)
    U2 = orc.updat(t, [X, X], [E, 2 * E], 1, alpha=0.5, beta=2.0, dw_in=U)
    np.testing.assert_allclose(U2, 0.5 * 3 * U + 2 * U, atol=1e-12)


def test_cfg0_random128(golden_dir):
    """BASELINE.json configs[0]: layout=random(128,128) bs 32 N=64 fp32 (host only)."""
    z = np.load(os.path.join(golden_dir, "cfg0_rand128.npz"))
    np.random.seed(0)
    layout = np.random.randint(2, size=(128, 128))
    t = orc.build_layout_luts(layout, 32)
    for axis in (0, 1):
        seed = int(z["a%d/seed" % axis])
        rng = np.random.RandomState(seed)
        f16 = lambda a: a.astype(np.float16).astype(np.float32)
        i_shape = (64, t["C"]) if axis else (t["C"], 64)
        o_shape = (64, t["K"]) if axis else (t["K"], 64)
        W = f16(rng.normal(0.0, 0.01, (t["blocks"], 32, 32)))
        X = f16(rng.normal(0.0, 0.1, i_shape))
        E = f16(rng.normal(0.0, 0.1, o_shape))
        for name, got, want in (("Y", orc.fprop_fast(t, X, W, axis, np.float64), z["a%d/Y" % axis]),
                                ("DX", orc.bprop_fast(t, E, W, axis, np.float64), z["a%d/DX" % axis]),
                                ("DW", orc.updat_fast(t, X, E, axis, np.float64)[::64], z["a%d/DW_every64" % axis])):
            scale = np.abs(want).max()
            assert np.abs(got - want).max() / scale < 1e-6, (axis, name)


def test_rounding_helpers():
    x = np.array([1.0, 1.00390625, 1.01171875, -3.14159, 65504.0, 1e-8], dtype=np.float32)
    b = orc.round_bf16(x)
    assert b[0] == 1.0 and b[1] == 1.0 and b[2] == np.float32(1.015625)   # ties-to-even both ways
    import torch
    tb = torch.tensor(x).to(torch.bfloat16).to(torch.float32).numpy()
    np.testing.assert_array_equal(b, tb)
    np.testing.assert_array_equal(orc.round_fp16(x), torch.tensor(x).to(torch.float16).to(torch.float32).numpy())


def test_identity_init_rule():
    lay = np.ones((3, 5), dtype=np.int32)
    t = orc.build_layout_luts(lay, 8)
    W = orc.identity_init(t, 2.0)
    for w, (c, k) in enumerate(t["updat_list"]):
        want = 2.0 * np.eye(8) if (c % 5) == (k % 3) else np.zeros((8, 8))
        np.testing.assert_array_equal(W[w], want)


def test_sampled_oracle_matches_the_full_oracle():
    """oracle.fprop_cols / bprop_rows / updat_blocks (used at BASELINE sizes) are the same sums as fprop / bprop / updat."""
    import _parity as P
    lay = P.random_layout(6, 9, 0.4, seed=3)
    for bs in (8, 32):
        t = orc.build_layout_luts(lay, bs)
        for axis in (0, 1):
            rng = np.random.default_rng(bs + axis)
            N = 20
            W = rng.normal(size=(t["blocks"], bs, bs))
            X = rng.normal(size=(N, 6 * bs) if axis else (6 * bs, N))
            E = rng.normal(size=(N, 9 * bs) if axis else (9 * bs, N))
            Y, DX, DW = orc.fprop(t, X, W, axis), orc.bprop(t, E, W, axis), orc.updat(t, X, E, axis)
            for k, v in orc.fprop_cols(t, X, W, axis, range(9)).items():
                np.testing.assert_allclose(Y[:, k * bs:(k + 1) * bs] if axis else Y[k * bs:(k + 1) * bs], v, atol=1e-12)
            for c, v in orc.bprop_rows(t, E, W, axis, range(6)).items():
                np.testing.assert_allclose(DX[:, c * bs:(c + 1) * bs] if axis else DX[c * bs:(c + 1) * bs], v, atol=1e-12)
            for w, v in orc.updat_blocks(t, X, E, axis, range(t["blocks"])).items():
                np.testing.assert_allclose(DW[w], v, atol=1e-12)
