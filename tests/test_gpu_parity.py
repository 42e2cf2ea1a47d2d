"""GPU parity: the HIP path (through the C ABI via blocksparse_amd.BlocksparseMatMul) vs the NumPy oracle,
mirroring the matrix of test/blocksparse_matmul_test.py (BA layout with locks, bsize 32/16/8, N sweep) and
extending it to both feature axes and fp32/fp16/bf16."""
import os

import numpy as np
import pytest

import _parity as P
from oracle import bsmm_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    from blocksparse_amd import BlocksparseMatMul, _lib
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    _lib.load()     # fail loudly if the HIP extension is missing
    return torch, BlocksparseMatMul, _lib


def _check(res, dtype, ctx):
    for name, (l2, mx) in res.items():
        assert l2 <= P.L2_BAR[dtype], "%s %s L2-rel %.3e > %.1e" % (ctx, name, l2, P.L2_BAR[dtype])
        assert mx <= P.MAX_BAR[dtype], "%s %s max-rel %.3e > %.1e" % (ctx, name, mx, P.MAX_BAR[dtype])


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("axis", [0, 1])
@pytest.mark.parametrize("bs", [32, 16, 8])
def test_ba_layout_all_passes(env, bs, axis, dtype):
    torch, BSMM, _ = env
    layout = P.ba_layout(40, 3, seed=1)
    for N in (64, 8):
        res = P.run_case(torch, BSMM, layout, bs, axis, dtype, N, seed=bs + axis)
        _check(res, dtype, "bs%d a%d %s N%d" % (bs, axis, dtype, N))


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("axis", [0, 1])
@pytest.mark.parametrize("bs", [32, 16, 8])
def test_ragged_minibatch_and_unaligned(env, bs, axis, dtype):
    """N not a multiple of the tile / vector width (reference needs N%4==0; we accept any N)."""
    torch, BSMM, _ = env
    layout = P.random_layout(6, 9, 0.4, seed=3)
    for N in (1, 5, 36, 100, 300):
        res = P.run_case(torch, BSMM, layout, bs, axis, dtype, N, seed=N)
        _check(res, dtype, "bs%d a%d %s N%d" % (bs, axis, dtype, N))


@pytest.mark.parametrize("axis", [0, 1])
@pytest.mark.parametrize("bs", [32, 16, 8])
def test_empty_rows_cols_and_single_block(env, bs, axis):
    torch, BSMM, _ = env
    layout = P.ba_layout(16, 2, seed=3)
    layout[:, 5] = 0
    layout[7, :] = 0
    _check(P.run_case(torch, BSMM, layout, bs, axis, "f32", 24), "f32", "holes")
    _check(P.run_case(torch, BSMM, np.ones((1, 1), dtype=np.int32), bs, axis, "f32", 8), "f32", "single")
    _check(P.run_case(torch, BSMM, np.ones((3, 2), dtype=np.int32), bs, axis, "bf16", 40), "bf16", "dense3x2")


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("axis", [0, 1])
@pytest.mark.parametrize("bs", [32, 16, 8])
def test_segmented_luts_with_locks(env, bs, axis, dtype):
    """Reference-policy tables (segments + lock ids) fed to the device: shared output blocks accumulate atomically (fp32)."""
    torch, BSMM, _ = env
    layout = P.ba_layout(64, 3, seed=5)
    b = BSMM(layout, block_size=bs, feature_axis=axis, segmented=True)
    assert b.fprop_locks > 0 and b.bprop_locks > 0
    res = P.run_case(torch, BSMM, layout, bs, axis, dtype, 48, seed=9, segmented=True, passes=("Y", "DX"))
    # shared output blocks are summed in fp32 (workspace image) and rounded once: the north-star bar holds for the
    # reference-format locked tables too (the reference itself rounds every partial sum to 16 bit)
    _check(res, dtype, "locked luts bs%d a%d" % (bs, axis))


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("bs,axis", [(32, 0), (32, 1), (16, 0), (16, 1), (8, 0), (8, 1)])
def test_mfma_and_valu_kernels_agree(env, bs, axis, dtype):
    """Two independent device implementations (matrix-core and plain VALU) on the same inputs."""
    torch, BSMM, lib = env
    layout = P.random_layout(10, 12, 0.35, seed=11)
    L = lib.load()
    try:
        lib.set_kernel_variant(1)
        a = P.run_case(torch, BSMM, layout, bs, axis, dtype, 72, seed=2)
    finally:
        lib.set_kernel_variant(0)
    b = P.run_case(torch, BSMM, layout, bs, axis, dtype, 72, seed=2)
    _check(a, dtype, "valu")
    _check(b, dtype, "mfma")


@pytest.mark.parametrize("bs", [32, 16])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("axis", [0, 1])
def test_plan_kernels_forced_on_small_and_ragged_cases(env, axis, dtype, bs):
    """The grouped (xcol) / windowed kernels normally run only when the problem fills the chip; force them
    (BSMM_FLAG_FORCE_PLAN on every call: lib.set_kernel_variant(3)) on small, ragged and degenerate cases: BA layout with hubs, empty rows/columns, odd
    block counts (partial groups, a trailing input pair without its odd block), single block, N not a multiple of the
    row tile, and N % 8 != 0 on axis 0 (must fall back to the generic kernel, still correct)."""
    torch, BSMM, lib = env
    L = lib.load()
    holes = P.ba_layout(16, 2, seed=3)
    holes[:, 5] = 0
    holes[7, :] = 0
    layouts = [P.ba_layout(40, 3, seed=1), holes, P.random_layout(7, 9, 0.5, seed=2), np.ones((1, 1), dtype=np.int32),
               P.random_layout(17, 33, 0.3, seed=6)]
    try:
        lib.set_kernel_variant(3)
        for li, layout in enumerate(layouts):
            for N in (8, 72, 200, 392) + ((100, 5) if li == 0 else ()):
                res = P.run_case(torch, BSMM, layout, bs, axis, dtype, N, seed=li * 10 + N)
                _check(res, dtype, "forced-plan bs%d layout%d a%d %s N%d" % (bs, li, axis, dtype, N))
    finally:
        lib.set_kernel_variant(0)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("axis", [0, 1])
def test_bsize8_super_block_path(env, axis, dtype):
    """bsize 8, 16-bit: the 'BSS8' plans run the bsize-32 matrix-core kernels on the 32x32 super-blocks (W expanded with
    zeros, DW gathered back).  Forced on small / ragged / degenerate cases like the other plan kernels; a 7x9 grid has no
    super plan (not a multiple of 4 blocks) and N % 8 != 0 on axis 0 must fall back to the V_FMA kernels, still correct."""
    torch, BSMM, lib = env
    L = lib.load()
    holes = P.ba_layout(16, 2, seed=3)
    holes[:, 5] = 0
    holes[7, :] = 0
    layouts = [P.ba_layout(40, 3, seed=1), holes, P.random_layout(8, 12, 0.5, seed=2), np.ones((4, 4), dtype=np.int32),
               P.random_layout(20, 36, 0.1, seed=6), P.random_layout(7, 9, 0.5, seed=2)]
    try:
        lib.set_kernel_variant(3)
        for li, layout in enumerate(layouts):
            for N in (8, 72, 200) + ((100, 5) if li == 0 else ()):
                res = P.run_case(torch, BSMM, layout, 8, axis, dtype, N, seed=li * 10 + N, segmented=(li == 4))
                _check(res, dtype, "super8 layout%d a%d %s N%d" % (li, axis, dtype, N))
    finally:
        lib.set_kernel_variant(0)


@pytest.mark.parametrize("axis", [0, 1])
def test_bsize8_super_block_updat_pairs_alpha_beta(env, axis):
    torch, BSMM, lib = env
    layout = P.random_layout(12, 8, 0.3, seed=8)
    b = BSMM(layout, block_size=8, feature_axis=axis)
    t = orc.build_layout_luts(layout, 8)
    N = 328
    Xs, Es = [], []
    for p in range(3):
        _, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), "bf16", seed=60 + p)
        Xs.append(X); Es.append(E)
    dw0 = orc.round_bf16(np.random.RandomState(1).normal(size=b.w_shape).astype(np.float32) * 0.1)
    ref = orc.updat(t, Xs, Es, axis, alpha=0.5, beta=2.0, dw_in=dw0)
    dw = P.to_dev(dw0, "bf16", torch)
    out = b.updat([P.to_dev(x, "bf16", torch) for x in Xs], [P.to_dev(e, "bf16", torch) for e in Es], alpha=0.5, beta=2.0, dw=dw)
    torch.cuda.synchronize()
    assert out.data_ptr() == dw.data_ptr()
    l2, mx = P.errors(P.to_host(out), orc.round_bf16(ref))
    assert l2 <= P.L2_BAR["bf16"], (l2, mx)     # fp32 sums -> alpha, beta -> ONE rounding, like the direct kernels


@pytest.mark.parametrize("axis", [0, 1])
def test_fp32_plan_kernels_forced(env, axis):
    """fp32 / bsize 32 has its own grouped kernel (the exact three-piece bf16 kernel of bsmm_xcols.h, both axes): same forced
    small / ragged cases, fp32 bar; N % 8 != 0 on axis 0 falls back to the generic kernel."""
    torch, BSMM, lib = env
    L = lib.load()
    holes = P.ba_layout(16, 2, seed=3)
    holes[:, 5] = 0
    holes[7, :] = 0
    layouts = [P.ba_layout(40, 3, seed=1), holes, P.random_layout(7, 9, 0.5, seed=2), np.ones((1, 1), dtype=np.int32)]
    try:
        lib.set_kernel_variant(3)
        for li, layout in enumerate(layouts):
            for N in (4, 72, 200, 392) + ((100, 5, 130) if li == 0 else ()):
                res = P.run_case(torch, BSMM, layout, 32, axis, "f32", N, seed=li * 10 + N)
                _check(res, "f32", "forced-plan f32 layout%d a%d N%d" % (li, axis, N))
    finally:
        lib.set_kernel_variant(0)


def test_fp32_split_kernel_is_exact_over_a_wide_exponent_range(env):
    """The axis-1 fp32 path multiplies bf16 PIECES (x = b1 + b2 + b3 exactly) on the 16-bit matrix cores.  Operands whose
    magnitudes span 2^-12 .. 2^12 element by element (so the pieces have very different scales) must still meet the fp32
    bar against the float64 oracle, in both passes, and agree with the fp32-MFMA per-segment kernels (variant 2)."""
    torch, BSMM, lib = env
    L = lib.load()
    layout = P.random_layout(12, 20, 0.4, seed=11)
    b = BSMM(layout, block_size=32, feature_axis=1)
    t = orc.build_layout_luts(layout, 32)
    N = 264
    rng = np.random.RandomState(5)
    def wide(shape, scale):
        return (rng.normal(size=shape) * scale * np.exp2(rng.randint(-12, 13, size=shape))).astype(np.float32)
    W, X, E = wide(b.w_shape, 0.01), wide(b.i_shape(N), 0.1), wide(b.o_shape(N), 0.1)
    w, x, e = (torch.from_numpy(a).cuda() for a in (W, X, E))
    try:
        lib.set_kernel_variant(3)
        y, dx = b.fprop(x, w), b.bprop(e, w)
        lib.set_kernel_variant(2)
        y2, dx2 = b.fprop(x, w), b.bprop(e, w)
        torch.cuda.synchronize()
    finally:
        lib.set_kernel_variant(0)
    for got, got2, ref, nm in ((y, y2, orc.fprop(t, X, W, 1), "Y"), (dx, dx2, orc.bprop(t, E, W, 1), "DX")):
        l2, mx = P.errors(P.to_host(got), ref)
        l2b, _ = P.errors(P.to_host(got2), ref)
        assert l2 <= P.L2_BAR["f32"], (nm, l2, mx)
        assert l2 <= 2 * l2b + 1e-8, (nm, l2, l2b)       # no worse than the fp32 matrix-core instruction


@pytest.mark.parametrize("axis", [0, 1])
def test_plan_and_generic_kernels_agree_at_scale(env, axis):
    """4096^2 / 20% at N = 2048 (large enough for the heuristic to pick the plan kernels): plan kernels vs the generic
    per-segment / per-block kernels (variant 2) on identical inputs."""
    torch, BSMM, lib = env
    L = lib.load()
    layout = P.random_layout(128, 128, 0.2, seed=1234)
    b = BSMM(layout, block_size=32, feature_axis=axis)
    g = torch.Generator(device="cuda").manual_seed(3)
    N = 2048
    w = (torch.randn(b.w_shape, device="cuda", generator=g) * 0.01).bfloat16()
    x = (torch.randn(b.i_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
    dy = (torch.randn(b.o_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
    outs = {}
    for v in (0, 2):
        lib.set_kernel_variant(v)
        try:
            outs[v] = (b.fprop(x, w).float(), b.bprop(dy, w).float(), b.updat(x, dy).float())
        finally:
            lib.set_kernel_variant(0)
    for name, p, q in zip(("Y", "DX", "DW"), outs[0], outs[2]):
        l2 = ((p - q).double().norm() / q.double().norm()).item()
        assert l2 < 2e-3, (name, l2)        # both are bf16-rounded results of fp32 sums in different orders


@pytest.mark.parametrize("split", ["1", "2", "4"])
@pytest.mark.parametrize("axis", [0, 1])
def test_windowed_updat_pairs_alpha_beta_and_minibatch_split(env, axis, split):
    """Windowed updat (bf16): 3 (x,dy) pairs, alpha/beta with DW accumulated in place (DWA form), and the
    split-minibatch path (fp32 atomics into the workspace + finalize kernel) forced via bsmm_args.split."""
    torch, BSMM, lib = env
    L = lib.load()
    layout = P.random_layout(12, 20, 0.4, seed=4)
    b = BSMM(layout, block_size=32, feature_axis=axis, updat_split=int(split))
    t = orc.build_layout_luts(layout, 32)
    N = 328
    Xs, Es = [], []
    for p in range(3):
        _, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), "bf16", seed=40 + p)
        Xs.append(X); Es.append(E)
    dw0 = orc.round_bf16(np.random.RandomState(1).normal(size=b.w_shape).astype(np.float32) * 0.1)
    ref = orc.updat(t, Xs, Es, axis, alpha=0.5, beta=2.0, dw_in=dw0)
    dw = P.to_dev(dw0, "bf16", torch)
    try:
        lib.set_kernel_variant(3)
        out = b.updat([P.to_dev(x, "bf16", torch) for x in Xs], [P.to_dev(e, "bf16", torch) for e in Es],
                      alpha=0.5, beta=2.0, dw=dw)
        torch.cuda.synchronize()
    finally:
        lib.set_kernel_variant(0)
    assert lib.last_kernel() == lib.K_UPDAT_STREAM        # (round 3: either feature axis)
    assert out.data_ptr() == dw.data_ptr()
    l2, mx = P.errors(P.to_host(out), orc.round_bf16(ref))
    assert l2 <= P.L2_BAR["bf16"], (l2, mx)


@pytest.mark.parametrize("axis", [0, 1])
@pytest.mark.parametrize("bs", [32, 8])
def test_updat_alpha_beta_and_pairs(env, bs, axis):
    torch, BSMM, _ = env
    layout = P.random_layout(5, 7, 0.5, seed=4)
    b = BSMM(layout, block_size=bs, feature_axis=axis)
    t = orc.build_layout_luts(layout, bs)
    N = 40
    Xs, Es = [], []
    for p in range(3):
        _, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), "f32", seed=20 + p)
        Xs.append(X); Es.append(E)
    dw0 = np.random.RandomState(0).normal(size=b.w_shape).astype(np.float32)
    ref = orc.updat(t, Xs, Es, axis, alpha=0.5, beta=2.0, dw_in=dw0)
    dw = P.to_dev(dw0, "f32", torch)
    out = b.updat([P.to_dev(x, "f32", torch) for x in Xs], [P.to_dev(e, "f32", torch) for e in Es],
                  alpha=0.5, beta=2.0, dw=dw)
    assert out.data_ptr() == dw.data_ptr()          # DWA form: accumulates in place
    l2, mx = P.errors(P.to_host(out), ref)
    assert l2 < 2e-6, l2


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("axis", [0, 1])
def test_autograd_matches_oracle(env, axis, dtype):
    torch, BSMM, _ = env
    layout = P.ba_layout(24, 3, seed=2)
    bs, N = 32, 64
    b = BSMM(layout, block_size=bs, feature_axis=axis)
    t = orc.build_layout_luts(layout, bs)
    W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), dtype, seed=5)
    w = P.to_dev(W, dtype, torch).requires_grad_()
    x = P.to_dev(X, dtype, torch).requires_grad_()
    y = b(x, w)
    y.backward(P.to_dev(E, dtype, torch))
    for name, got, ref in (("Y", y, orc.fprop(t, X, W, axis)), ("DX", x.grad, orc.bprop(t, E, W, axis)),
                           ("DW", w.grad, orc.updat(t, X, E, axis))):
        l2, mx = P.errors(P.to_host(got), orc.round_to(ref, dtype))
        assert l2 <= P.L2_BAR[dtype], (name, l2)


@pytest.mark.parametrize("dtype,bs,axis", [("f32", 16, 0), ("bf16", 32, 1)])
def test_grouped_dw_over_many_pairs(env, dtype, bs, axis):
    """19 (x, dy) pairs through chained DW / DWA launches of <= 8 pairs (the group_param_grads pattern)."""
    torch, BSMM, _ = env
    layout = P.random_layout(6, 5, 0.5, seed=9)
    b = BSMM(layout, block_size=bs, feature_axis=axis)
    t = orc.build_layout_luts(layout, bs)
    N = 40
    Xs, Es = [], []
    for p in range(19):
        _, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), dtype, seed=60 + p)
        Xs.append(X); Es.append(E)
    ref = orc.updat(t, Xs, Es, axis)
    out = b.updat_grouped([P.to_dev(x, dtype, torch) for x in Xs], [P.to_dev(e, dtype, torch) for e in Es], group_size=8)
    l2, _ = P.errors(P.to_host(out), orc.round_to(ref, dtype))
    assert l2 <= (2e-6 if dtype == "f32" else 4e-3), l2     # bf16: three roundings of the running sum (DWA stores bf16)


def test_rank3_inputs_flatten_non_feature_dims(env):
    torch, BSMM, _ = env
    layout = P.random_layout(4, 4, 0.6, seed=8)
    b1 = BSMM(layout, block_size=16, feature_axis=1)
    w = torch.randn(b1.w_shape, device="cuda", generator=P.gen(torch, 31)) * 0.05
    x = torch.randn(3, 5, b1.C, device="cuda", generator=P.gen(torch, 32))
    y = b1(x, w)
    assert tuple(y.shape) == (3, 5, b1.K)
    torch.testing.assert_close(y.reshape(15, -1), b1(x.reshape(15, -1), w))
    b0 = BSMM(layout, block_size=16, feature_axis=0)
    x0 = torch.randn(b0.C, 3, 4, device="cuda", generator=P.gen(torch, 33))
    y0 = b0(x0, w)
    assert tuple(y0.shape) == (b0.K, 3, 4)
    torch.testing.assert_close(y0.reshape(b0.K, 12), b0(x0.reshape(b0.C, 12), w))


def test_identity_init_and_one_mode(env):
    """The reference's `one=1` debug mode (test/blocksparse_matmul_test.py:301-307,330-332): all-ones X with
    identity blocks gives Y[k,:] = number of blocks in column k."""
    torch, BSMM, _ = env
    layout = P.ba_layout(20, 2, seed=7)
    for bs, axis in ((32, 0), (16, 1), (8, 0)):
        b = BSMM(layout, block_size=bs, feature_axis=axis)
        t = orc.build_layout_luts(layout, bs)
        W = b.identity_init(2.0)()
        np.testing.assert_array_equal(P.to_host(W), orc.identity_init(t, 2.0))
        Wi = torch.eye(bs, device="cuda").repeat(b.blocks, 1, 1)
        x = torch.ones(b.i_shape(16), device="cuda")
        y = P.to_host(b.fprop(x, Wi))
        counts = layout.sum(axis=0)
        want = np.repeat(counts, bs)[:, None] * np.ones((1, 16)) if axis == 0 else np.ones((16, 1)) * np.repeat(counts, bs)[None, :]
        np.testing.assert_array_equal(y, want)


def test_cfg0_golden_on_device(env, golden_dir):
    """BASELINE.json configs[0] (layout=random(128,128), bs 32, N=64, fp32) against the REFERENCE's own outputs."""
    torch, BSMM, _ = env
    z = np.load(os.path.join(golden_dir, "cfg0_rand128.npz"))
    np.random.seed(0)
    layout = np.random.randint(2, size=(128, 128))
    for axis in (0, 1):
        b = BSMM(layout, block_size=32, feature_axis=axis)
        rng = np.random.RandomState(int(z["a%d/seed" % axis]))
        f16 = lambda a: a.astype(np.float16).astype(np.float32)
        W = f16(rng.normal(0.0, 0.01, b.w_shape))
        X = f16(rng.normal(0.0, 0.1, b.i_shape(64)))
        E = f16(rng.normal(0.0, 0.1, b.o_shape(64)))
        w, x, e = (P.to_dev(a, "f32", torch) for a in (W, X, E))
        for name, got, ref in (("Y", b.fprop(x, w), z["a%d/Y" % axis]), ("DX", b.bprop(e, w), z["a%d/DX" % axis]),
                               ("DW", b.updat(x, e)[::64], z["a%d/DW_every64" % axis])):
            l2, mx = P.errors(P.to_host(got), ref)
            assert l2 < 2e-6, (axis, name, l2)


def test_golden_math_fixtures_on_device(env, golden_dir):
    """Every small case of tests/golden/math.npz (outputs of the reference's fprop_test/bprop_test/updat_test)."""
    torch, BSMM, _ = env
    z = np.load(os.path.join(golden_dir, "math.npz"))
    for lay in ("ba16", "holes", "rect", "single"):
        for bs in (8, 16, 32):
            for axis in (0, 1):
                g = lambda k: z["%s/bs%d/a%d/%s" % (lay, bs, axis, k)]
                b = BSMM(g("layout"), block_size=bs, feature_axis=axis)
                w, x, e = (P.to_dev(g(k).astype(np.float32), "f32", torch) for k in ("W", "X", "E"))
                for name, got in (("Y", b.fprop(x, w)), ("DX", b.bprop(e, w)), ("DW", b.updat(x, e))):
                    l2, mx = P.errors(P.to_host(got), g(name))
                    assert l2 < 2e-6, (lay, bs, axis, name, l2)


@pytest.mark.parametrize("dtype,bs,axis", [("f32", 32, 1), ("bf16", 32, 1), ("bf16", 32, 0), ("bf16", 16, 0), ("bf16", 8, 0)])
def test_full_size_properties(env, dtype, bs, axis):
    """BASELINE-size layouts (4096^2) checked through size-independent properties:
    (1) linearity  f(x1 + x2) == f(x1) + f(x2) on inputs whose sums are exactly representable,
    (2) <dy, fprop(x)> == <bprop(dy), x> == <updat(x, dy), w>  (adjointness of the three passes),
    (3) a sampled set of output blocks against the oracle."""
    torch, BSMM, _ = env
    CB = 4096 // bs
    layout = P.random_layout(CB, CB, 0.2 if bs == 32 else 0.1, seed=1234)
    N = 512
    b = BSMM(layout, block_size=bs, feature_axis=axis)
    td = getattr(torch, P.TORCH_DT[dtype])
    g = torch.Generator(device="cuda").manual_seed(1)
    # small integers / 64: sums and products stay exact enough for tight comparisons
    w = (torch.randint(-4, 5, b.w_shape, device="cuda", generator=g).float() / 64).to(td)
    x1 = (torch.randint(-4, 5, b.i_shape(N), device="cuda", generator=g).float() / 8).to(td)
    x2 = (torch.randint(-4, 5, b.i_shape(N), device="cuda", generator=g).float() / 8).to(td)
    dy = (torch.randint(-4, 5, b.o_shape(N), device="cuda", generator=g).float() / 8).to(td)
    y1, y2, y12 = b.fprop(x1, w).float(), b.fprop(x2, w).float(), b.fprop(x1 + x2, w).float()
    tol = 1e-5 if dtype == "f32" else 2e-2
    assert (y12 - (y1 + y2)).abs().max().item() <= tol * max(1.0, y12.abs().max().item())
    dx = b.bprop(dy, w).float()
    dw = b.updat(x1, dy).float()
    s1 = (dy.double() * y1.double()).sum().item()
    s2 = (dx.double() * x1.double()).sum().item()
    s3 = (dw.double() * w.double()).sum().item()
    # the three inner products differ only by the storage rounding of y / dx / dw (<= 2^-9 relative per element
    # in bf16): compare against the noise scale sqrt(sum (a_i b_i)^2), not against the (cancelling) sum itself
    noise = ((dy.double() * y1.double()).square().sum().sqrt() + (dx.double() * x1.double()).square().sum().sqrt()
             + (dw.double() * w.double()).square().sum().sqrt()).item()
    rel = 1e-5 if dtype == "f32" else 2.0 ** -7
    assert abs(s1 - s2) <= rel * noise and abs(s1 - s3) <= rel * noise, (s1, s2, s3, noise)
    # sampled output blocks vs oracle
    t = orc.build_layout_luts(layout, bs)
    X = P.to_host(x1); Wn = P.to_host(w)
    Y = P.to_host(y1)
    cols = dict(t["fprop_list"])
    for k in (0, CB // 3, CB - 1):
        ref = np.zeros((bs, N)) if axis == 0 else np.zeros((N, bs))
        for c, wi in cols[k]:
            if axis == 0:
                ref += Wn[wi].astype(np.float64).T @ X[c * bs:(c + 1) * bs, :]
            else:
                ref += X[:, c * bs:(c + 1) * bs].astype(np.float64) @ Wn[wi]
        got = Y[k * bs:(k + 1) * bs, :] if axis == 0 else Y[:, k * bs:(k + 1) * bs]
        l2, _ = P.errors(got, orc.round_to(ref, dtype))
        assert l2 <= P.L2_BAR[dtype], (k, l2)


def test_step_is_capturable_in_a_hip_graph(env):
    """The C-ABI launches go to torch's current stream with no host synchronisation or allocation of their own, so a whole
    fprop + updat + bprop step can be captured once (torch.cuda.CUDAGraph = hipGraph) and replayed; results are bitwise
    those of the eager launches."""
    torch, BSMM, lib = env
    b = BSMM(P.random_layout(16, 16, 0.3, seed=4), block_size=32, feature_axis=1)
    N = 96
    g0 = torch.Generator(device="cuda").manual_seed(5)
    w = (torch.randn(b.w_shape, device="cuda", generator=g0) * 0.01).bfloat16()
    x = (torch.randn(b.i_shape(N), device="cuda", generator=g0) * 0.1).bfloat16()
    dy = (torch.randn(b.o_shape(N), device="cuda", generator=g0) * 0.1).bfloat16()
    dw = torch.empty(b.w_shape, dtype=torch.bfloat16, device="cuda")

    def step():
        y = b.fprop(x, w)
        b.updat(x, dy, dw=dw)
        return y, b.bprop(dy, w)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()                                   # warm-up outside the capture (function attributes, table uploads)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        y_g, dx_g = step()
    x.mul_(0.5)                                      # new inputs in the captured buffers
    graph.replay()
    torch.cuda.synchronize()
    dw_g = dw.clone()
    y_e, dx_e = step()
    torch.cuda.synchronize()
    assert torch.equal(y_g, y_e) and torch.equal(dx_g, dx_e) and torch.equal(dw_g, dw)


@pytest.mark.parametrize("axis", [0, 1])
@pytest.mark.parametrize("bs", [32, 16, 8])
def test_inf_in_an_unconnected_feature_does_not_reach_the_output(env, bs, axis):
    """The reference never reads an input feature block that the layout does not connect to an output block
    (blocksparse/matmul.py:353-392 walks only the lut entries): an Inf / NaN there leaves that output finite.  The plan kernels
    that K-concatenate blocks (bsize 16: two input blocks per v_mfma_f32_16x16x32) mask the ACTIVATIONS of an absent partner,
    not only its weights, so the same holds for them -- forced here with BSMM_FLAG_FORCE_PLAN.  bsize 8 with a plan multiplies
    zero-filled 32x32 super-blocks, where 0 * Inf WOULD appear: the call scans the activations for non-finite values and, when
    there is one, recomputes the output with the per-entry kernel after the matrix-core pass (round 4; bsmm_super8.h) -- checked
    here on the super-block path itself (FORCE_PLAN, BSMM_K_XPROP_SUPER8 asserted), with an Inf, with a NaN, and with clean inputs
    (whose result must be the matrix-core one: identical to a second clean call)."""
    torch, BSMM, lib = env
    rng = np.random.default_rng(3 + bs + axis)
    CB = KB = 48 if bs != 8 else 64
    lay = (rng.random((CB, KB)) < 0.12).astype(np.int32)
    lay[np.arange(CB), rng.integers(0, KB, CB)] = 1
    lay[rng.integers(0, CB, KB), np.arange(KB)] = 1
    N = 256
    b = BSMM(lay, block_size=bs, feature_axis=axis)
    g = torch.Generator(device="cuda").manual_seed(1)
    w = (torch.randn(b.w_shape, device="cuda", generator=g) * 0.05).bfloat16()
    x = (torch.randn(b.i_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
    dy = (torch.randn(b.o_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
    c_bad, k_bad = 5, 7
    xs = x.clone()
    (xs[:, c_bad * bs:(c_bad + 1) * bs] if axis else xs[c_bad * bs:(c_bad + 1) * bs, :]).fill_(float("inf"))
    es = dy.clone()
    (es[:, k_bad * bs:(k_bad + 1) * bs] if axis else es[k_bad * bs:(k_bad + 1) * bs, :]).fill_(float("nan"))
    lib.set_kernel_variant(3)
    try:
        y = b.fprop(xs, w).float()
        k_f = lib.last_kernel()
        dx = b.bprop(es, w).float()
        assert lib.last_kernel() == k_f and k_f in ((lib.K_XPROP_SUPER8,) if bs == 8 else (lib.K_XCOL32_STAGED, lib.K_XCOL32_FLOW, lib.K_XCOL16_STAGED))
        if bs == 8:
            # where the output is finite it must also be RIGHT (the repair pass rewrote all of it): against the exact kernels
            lib.set_kernel_variant(2)
            y_e, dx_e = b.fprop(xs, w).float(), b.bprop(es, w).float()
            lib.set_kernel_variant(3)
            fin = torch.isfinite(y_e)
            assert torch.equal(torch.isfinite(y), fin) and torch.allclose(y[fin], y_e[fin], rtol=2e-2, atol=1e-3)
            fin = torch.isfinite(dx_e)
            assert torch.equal(torch.isfinite(dx), fin) and torch.allclose(dx[fin], dx_e[fin], rtol=2e-2, atol=1e-3)
            # clean inputs: the flag stays clear, the matrix-core result stands (and a dirty call in between does not stick)
            y_c1 = b.fprop(x, w); b.fprop(xs, w); y_c2 = b.fprop(x, w)
            assert lib.last_kernel() == lib.K_XPROP_SUPER8 and torch.equal(y_c1, y_c2) and bool(torch.isfinite(y_c1.float()).all())
    finally:
        lib.set_kernel_variant(0)
    for k in range(KB):
        blk = y[:, k * bs:(k + 1) * bs] if axis else y[k * bs:(k + 1) * bs, :]
        assert bool(torch.isfinite(blk).all()) == (lay[c_bad, k] == 0), ("fprop", bs, axis, k)
    for c in range(CB):
        blk = dx[:, c * bs:(c + 1) * bs] if axis else dx[c * bs:(c + 1) * bs, :]
        assert bool(torch.isfinite(blk).all()) == (lay[c, k_bad] == 0), ("bprop", bs, axis, c)


def test_prepared_weights_cache_fp32(env):
    """bsmm_prepare_weights: the bf16 pieces of constant fp32 weights are made once per parameter version; fprop / bprop with the
    cached pieces give the same bits as a call that splits W itself, and an in-place update of W refreshes them."""
    import ctypes
    torch, BSMM, lib = env
    lay = P.random_layout(40, 40, 0.2, seed=5)
    b = BSMM(lay, block_size=32, feature_axis=1)
    N = 4096
    g = torch.Generator(device="cuda").manual_seed(2)
    w = torch.randn(b.w_shape, device="cuda", generator=g) * 0.05
    x = torch.randn(b.i_shape(N), device="cuda", generator=g) * 0.1
    dy = torch.randn(b.o_shape(N), device="cuda", generator=g) * 0.1
    lib.set_kernel_variant(3)
    try:
        y1, dx1 = b.fprop(x, w), b.bprop(dy, w)
        assert lib.last_kernel() == lib.K_XCOL32_F32SPLIT and set(b._prepared_w) == {lib.OP_FPROP, lib.OP_BPROP}
        keys = {k: v[1] for k, v in b._prepared_w.items()}
        y2 = b.fprop(x, w)
        assert torch.equal(y1, y2) and b._prepared_w[lib.OP_FPROP][1] == keys[lib.OP_FPROP]            # cache hit
        # the uncached call (prepared_w = NULL: the library splits W into the workspace) gives the same bits
        L = lib.load()
        tabs = b._tables_on(x.device)
        a = b._args(tabs.fprop, b._dev_tables["fprop"], N, b.C, b.K, x.dtype, plan=tabs.fprop_plan_f32)
        ws = torch.empty(L.bsmm_workspace_bytes(lib.OP_FPROP, ctypes.byref(a)), dtype=torch.uint8, device="cuda")
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        y3 = torch.empty_like(y1)
        assert L.bsmm_fprop(x.data_ptr(), w.data_ptr(), y3.data_ptr(), ctypes.byref(a)) == 0
        assert torch.equal(y1, y3)
        w.mul_(2.0)                                                                                   # optimizer step: version changes
        y4 = b.fprop(x, w)
        assert b._prepared_w[lib.OP_FPROP][1] != keys[lib.OP_FPROP]
        assert torch.allclose(y4, 2.0 * y1, rtol=1e-6, atol=1e-6)
        dx2 = b.bprop(dy, w)
        assert torch.allclose(dx2, 2.0 * dx1, rtol=1e-6, atol=1e-6)
        # a temporary at a recycled address with the same version is ANOTHER tensor, not a cache hit
        wa = (w * 0.5).contiguous()
        ya = b.fprop(x, wa)
        ptr, ver = wa.data_ptr(), wa._version
        del wa
        wb = (w * 0.25).contiguous()                                                                  # same size: usually the same block
        recycled = wb.data_ptr() == ptr and wb._version == ver
        yb = b.fprop(x, wb)
        assert torch.allclose(yb, 0.5 * ya, rtol=1e-6, atol=1e-6), ("stale prepared weights", recycled)
        # a mutation the version counter cannot see needs invalidate_weights(); cache_prepared = False never caches
        wb.data.mul_(2.0)
        b.invalidate_weights()
        assert torch.allclose(b.fprop(x, wb), ya, rtol=1e-6, atol=1e-6)
        b.cache_prepared = False
        wb.data.mul_(2.0)
        assert torch.allclose(b.fprop(x, wb), 2.0 * ya, rtol=1e-6, atol=1e-6)
        b.cache_prepared = True
    finally:
        lib.set_kernel_variant(0)
