"""GPU parity at the BASELINE.json configurations themselves (not scaled-down stand-ins), production dispatch
(no kernel-choice flags unless a case says so), against the float64 oracle on sampled output blocks; every case asserts
through bsmm_args.trace WHICH kernel family ran, so a silent fall-back to a generic kernel fails the test.

  (a) the bench shape: 4096^2, bsize 32, bf16, minibatch 8192, density 10 / 20 / 50 % (feature axis 1; 20 % on axis 0 too)
  (b) BASELINE configs[3]: 8192^2, 5 %, bf16, the per-GPU shard N = 512 and the whole minibatch N = 4096, all three passes
  (c) forced window sides of the streaming updat plan (BSMM_PLAN_STREAM_* and the legacy BSMM_PLAN_WINDOW_* names)
  (d) small forced-plan layouts whose 16x16 windows hold several blocks (multi-block windows, shared fragments, split items)
  (e) the reference's own test matrix at its own size: Barabasi-Albert(160, 5) + I, N in {256,128,64,32,16,8},
      bsize 32/16/8 (test/blocksparse_matmul_test.py:276-328), here on both feature axes
  (f) BASELINE configs[2]: 4096^2, bsize 16, 10 %, bf16, axis 0, minibatch 8192
"""
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
    _lib.load()
    return torch, BlocksparseMatMul, _lib


def _updat_kernel(lib, axis, opt=0):
    """which bsize-32 updat kernel a plan built with `opt` runs: the streaming kernel (bsmm_updat_v2.h) on both axes -- the windowed
    kernels of round 1 were retired in round 4, BSMM_PLAN_WINDOW_* now name the window side of the streaming plan"""
    return lib.K_UPDAT_STREAM


def _xprop_kernel(lib, axis, opt=0):
    """which bsize-32 16-bit xprop plan kernel an UNGATED call runs: feature axis 1 -> the barrier-free persistent kernel
    (bsmm_xflow.h), feature axis 0 -> the staged kernel (bsmm_xcol_v2.h), unless the plan options name the round-1 kernel"""
    if opt & (lib.PLAN_XCOL_UNSTAGED | lib.PLAN_XCOL_NARROW):
        return lib.K_XCOL32
    return lib.K_XCOL32_FLOW if axis == 1 else lib.K_XCOL32_STAGED


def _inputs(torch, b, N, dtype, seed):
    """W ~ N(0, .01), X, E ~ N(0, .1) generated on the device, rounded to the storage type; host copies are exact."""
    td = getattr(torch, P.TORCH_DT[dtype])
    g = torch.Generator(device="cuda").manual_seed(seed)
    w = (torch.randn(b.w_shape, device="cuda", generator=g) * 0.01).to(td)
    x = (torch.randn(b.i_shape(N), device="cuda", generator=g) * 0.1).to(td)
    e = (torch.randn(b.o_shape(N), device="cuda", generator=g) * 0.1).to(td)
    return w, x, e


def _spread(n, count):
    """`count` indices spread over range(n), hitting different residues mod 16 (= different waves of a workgroup)."""
    if n <= count:
        return list(range(n))
    step = n / float(count)
    return sorted(set(min(n - 1, int(i * step) + (i % 16 if int(i * step) + (i % 16) < n else 0)) for i in range(count)) | {0, n - 1})


def _check_sampled(torch, lib, b, layout, N, dtype, seed, expect, ctx, n_cols=12, n_blocks=96, passes=("Y", "DX", "DW")):
    """fprop / bprop / updat on the device; sampled output block columns / rows / weight blocks vs the float64 oracle."""
    bs, axis = b.bsize, b.axis
    t = orc.build_layout_luts(layout, bs)
    w, x, e = _inputs(torch, b, N, dtype, seed)
    W, X, E = P.to_host(w), P.to_host(x), P.to_host(e)
    bar = P.L2_BAR[dtype]

    def blk(a, i):
        return a[:, i * bs:(i + 1) * bs] if axis else a[i * bs:(i + 1) * bs, :]

    if "Y" in passes:
        y = P.to_host(b.fprop(x, w))
        if expect.get("xprop") is not None:
            assert lib.last_kernel() == expect["xprop"], "%s fprop ran kernel %d" % (ctx, lib.last_kernel())
        assert np.isfinite(y).all()
        for k, ref in orc.fprop_cols(t, X, W, axis, _spread(b.KB, n_cols)).items():
            l2, mx = P.errors(blk(y, k), orc.round_to(ref, dtype))
            assert l2 <= bar, "%s Y col %d L2 %.3e" % (ctx, k, l2)
    if "DX" in passes:
        dx = P.to_host(b.bprop(e, w))
        if expect.get("xprop") is not None:
            assert lib.last_kernel() == expect["xprop"], "%s bprop ran kernel %d" % (ctx, lib.last_kernel())
        assert np.isfinite(dx).all()
        for c, ref in orc.bprop_rows(t, E, W, axis, _spread(b.CB, n_cols)).items():
            l2, mx = P.errors(blk(dx, c), orc.round_to(ref, dtype))
            assert l2 <= bar, "%s DX row %d L2 %.3e" % (ctx, c, l2)
    if "DW" in passes:
        dw = P.to_host(b.updat(x, e))
        if expect.get("updat") is not None:
            assert lib.last_kernel() == expect["updat"], "%s updat ran kernel %d" % (ctx, lib.last_kernel())
        assert np.isfinite(dw).all()
        ws = _spread(b.blocks, n_blocks)
        ref = orc.updat_blocks(t, X, E, axis, ws)
        got = np.stack([dw[i] for i in ws])
        want = np.stack([orc.round_to(ref[i], dtype) for i in ws])
        l2, mx = P.errors(got, want)
        assert l2 <= bar, "%s DW sampled L2 %.3e" % (ctx, l2)
        P.assert_blocks(got, np.stack([ref[i] for i in ws]), dtype, len(ws), ctx + " DW sampled blocks")
        # every block was written (an item the plan forgot would leave torch.empty garbage or zeros): column sums of |dw|
        assert (np.abs(dw).reshape(b.blocks, -1).sum(axis=1) > 0).all(), ctx


# ---- (a) the bench shape ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("density,axis", [(0.1, 1), (0.2, 1), (0.5, 1), (0.2, 0)])
def test_bench_shape_against_oracle(env, density, axis):
    torch, BSMM, lib = env
    layout = P.random_layout(128, 128, density, seed=1234)          # bench.py's layout
    b = BSMM(layout, block_size=32, feature_axis=axis)
    _check_sampled(torch, lib, b, layout, 8192, "bf16", seed=11, expect={"xprop": _xprop_kernel(lib, axis), "updat": _updat_kernel(lib, axis)},
                   ctx="bench d%d a%d" % (round(density * 100), axis))


def test_bench_shape_fp16_and_ragged_minibatch(env):
    """fp16 storage and a minibatch that is not a multiple of the row tile (8192 - 24): last row tile / last chunk partial."""
    torch, BSMM, lib = env
    layout = P.random_layout(128, 128, 0.2, seed=1234)
    b = BSMM(layout, block_size=32, feature_axis=1)
    _check_sampled(torch, lib, b, layout, 8192 - 24, "f16", seed=12, expect={"xprop": lib.K_XCOL32_FLOW, "updat": lib.K_UPDAT_STREAM}, ctx="bench f16 ragged")


def test_bench_shape_skewed_layout(env):
    """Barabasi-Albert + I layout of the bench size (hub rows / columns with ~100 blocks, most windows nearly empty): the
    reference's own bench layout (test/blocksparse_matmul_bench.py:66-68)."""
    torch, BSMM, lib = env
    layout = P.ba_layout(128, 14, seed=1)
    b = BSMM(layout, block_size=32, feature_axis=1)
    _check_sampled(torch, lib, b, layout, 4096, "bf16", seed=13, expect={"xprop": lib.K_XCOL32_FLOW, "updat": lib.K_UPDAT_STREAM}, ctx="bench BA")


@pytest.mark.parametrize("bs,axis,density", [(32, 1, 0.2), (32, 0, 0.2), (16, 0, 0.1), (16, 1, 0.1), (8, 0, 0.1), (8, 1, 0.1)])
def test_full_output_at_4096_against_float64_oracle(env, bs, axis, density):
    """EVERY element of Y, DX and DW at 4096^2 / N = 1024 under FORCE_PLAN against the float64 oracle (`*_fast` evaluated in
    float64 on the rounded inputs, rounded once to bf16): a plan that mis-schedules one output column, one window or one block
    anywhere fails here -- the sampled checks at N = 8192 above cannot see that.  Per-column / per-block
    L2 errors, so a single bad column is not averaged away."""
    torch, BSMM, lib = env
    CB = 4096 // bs
    layout = P.random_layout(CB, CB, density, seed=1234)
    b = BSMM(layout, block_size=bs, feature_axis=axis)
    N = 1024
    w, x, e = _inputs(torch, b, N, "bf16", seed=21)
    W, X, E = P.to_host(w), P.to_host(x), P.to_host(e)
    t = orc.build_layout_luts(layout, bs)
    bar = P.L2_BAR["bf16"]
    want = {8: (lib.K_XPROP_SUPER8, lib.K_UPDAT_SUPER8), 16: (lib.K_XCOL16_STAGED, lib.K_UPDAT16_ROWS if axis == 0 else lib.K_UPDAT16_WIN), 32: (lib.K_XCOL32_FLOW if axis == 1 else lib.K_XCOL32_STAGED, lib.K_UPDAT_STREAM)}[bs]
    lib.set_kernel_variant(3)
    try:
        y = P.to_host(b.fprop(x, w)); kf = lib.last_kernel()
        dx = P.to_host(b.bprop(e, w)); kb = lib.last_kernel()
        dw = P.to_host(b.updat(x, e)); ku = lib.last_kernel()
    finally:
        lib.set_kernel_variant(0)
    assert (kf, kb, ku) == (want[0], want[0], want[1]), (kf, kb, ku)

    # one criterion for every grouping (tests/_parity.py::assert_blocks): element within one storage-type step, block <= 4 x bar, tensor <= bar
    for name, got, ref, nb in (("Y", y, orc.fprop_fast(t, X, W, axis, dtype=np.float64), b.KB),
                               ("DX", dx, orc.bprop_fast(t, E, W, axis, dtype=np.float64), b.CB)):
        P.assert_blocks(P.act_blocks(got, axis, N, nb, bs), P.act_blocks(ref, axis, N, nb, bs), "bf16", nb, (name, bs, axis))
    P.assert_blocks(dw, orc.updat_fast(t, X, E, axis, dtype=np.float64), "bf16", b.blocks, ("DW", bs, axis))


@pytest.mark.parametrize("bsize,axis,kernel", [(32, 1, "K_UPDAT_STREAM"), (16, 1, "K_UPDAT16_WIN"), (8, 1, "K_UPDAT_SUPER8"), (8, 0, "K_UPDAT_SUPER8")])
def test_fp32_updat_split_with_huge_and_nonfinite_inputs(env, bsize, axis, kernel):
    """The fp32 weight gradient through the bf16 three-piece split (bsize 32 / 16 on feature axis 1, bsize 8 on either) with activations the
    FIRST piece cannot hold:
    (a) |x| above the largest bf16 (3.39e38 < |x| <= FLT_MAX) used to round to Inf and leave NaN pieces -- the first piece is clamped, the
        split stays exact and the blocks come out as the float64 oracle's;
    (b) an Inf or NaN activation cannot be split: the call raises its flag and the per-block fp32 kernel computes it (round 5) -- the
        result then has Inf and NaN exactly where IEEE arithmetic on the unsplit values has them (the float64 oracle's), the finite rest
        within the fp32 bar, with alpha / beta applied ONCE (the skipped finalize pass must not have touched DW)."""
    torch, BSMM, lib = env
    nb = 24 * 32 // bsize
    lay = P.random_layout(nb, nb, 0.3 if bsize == 32 else 0.2, seed=4)
    N = 512
    b = BSMM(lay, block_size=bsize, feature_axis=axis)
    t = orc.build_layout_luts(np.asarray(lay), bsize)
    W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), "f32", seed=41)
    E = (E * 1e-3).astype(np.float32)
    big = np.float32(3.4e38)                                   # finite in fp32, beyond the bf16 range
    lut = np.asarray(b.updat_lut).reshape(-1, 2)
    c0 = int(lut[0, 0])

    def put(A, n, f, v):                                       # element (minibatch n, feature f) in the layout of the feature axis
        if axis == 1: A[n, f] = v
        else:         A[f, n] = v
    put(X, 3, bsize * c0 + 5, big)
    put(X, 7, bsize * c0 + 6, -big)
    x, e = P.to_dev(X, "f32", torch), P.to_dev(E, "f32", torch)
    got = P.to_host(b.updat(x, e))
    assert lib.last_kernel() == getattr(lib, kernel)
    ref = orc.updat(t, X.astype(np.float64), E.astype(np.float64), axis)
    assert np.isfinite(got).all()
    l2, _ = P.errors(got, ref)
    assert l2 <= P.L2_BAR["f32"], l2
    # (b) an Inf in X, a NaN in DY, accumulated into an existing DW
    put(X, 3, bsize * c0 + 5, np.inf)
    k0 = int(lut[-1, 1])
    put(E, 11, bsize * k0 + 2, np.nan)
    dw0 = (np.random.default_rng(5).standard_normal(b.w_shape) * 0.1).astype(np.float32)
    dw = P.to_dev(dw0.copy(), "f32", torch)
    got2 = P.to_host(b.updat(P.to_dev(X, "f32", torch), P.to_dev(E, "f32", torch), alpha=0.5, beta=2.0, dw=dw))
    assert lib.last_kernel() == getattr(lib, kernel)           # (the trace names the path the call took; the repair pass rides behind it)
    with np.errstate(invalid="ignore", over="ignore"):
        ref2 = 0.5 * orc.updat(t, X.astype(np.float64), E.astype(np.float64), axis) + 2.0 * dw0.astype(np.float64)
    assert np.isnan(ref2).any() and np.isinf(ref2).any()
    assert np.array_equal(np.isnan(got2), np.isnan(ref2))
    inf = np.isinf(ref2)
    assert np.array_equal(np.isinf(got2), inf) and np.array_equal(np.sign(got2[inf]), np.sign(ref2[inf]))
    fin = np.isfinite(ref2)
    l2, _ = P.errors(np.where(fin, got2, 0), np.where(fin, ref2, 0))
    assert l2 <= P.L2_BAR["f32"], l2
    # ... and the next finite call is back on the split path's result, bit for bit
    assert np.array_equal(P.to_host(b.updat(x, e)), got)


@pytest.mark.parametrize("density", [0.1, 0.2, 0.5])
def test_timed_kernel_variants_full_output_at_n8192(env, density):
    """The kernels bench.py TIMES (4096^2, bs 32, feature axis 1, N = 8192: the flow kernel with 128-row units, the streaming weight
    gradient), every output element (VERDICT r4, weak 4 -- the N = 1024 test above runs the flow kernel's 64-row variant):
    (a) fprop / bprop of the flow kernel, 128-row units asserted through the trace's variant byte, bit-equal to the staged kernel
        (which test_full_output_at_4096... holds against the float64 oracle);
    (b) every block of DW from the streaming kernel against the float64 oracle, per block.
    Reference counterpart: the full-tensor comparisons of test/blocksparse_matmul_test.py:396-421."""
    torch, BSMM, lib = env
    layout = P.random_layout(128, 128, density, seed=1234)
    b = BSMM(layout, block_size=32, feature_axis=1)
    bs_ = BSMM(layout, block_size=32, feature_axis=1)
    bs_.flow = False
    N = 8192
    w, x, e = _inputs(torch, b, N, "bf16", seed=33)
    y = b.fprop(x, w); kf, vf = lib.last_kernel(), lib.last_kernel_variant()
    dx = b.bprop(e, w); kb, vb = lib.last_kernel(), lib.last_kernel_variant()
    assert (kf, vf, kb, vb) == (lib.K_XCOL32_FLOW, 0, lib.K_XCOL32_FLOW, 0), (kf, vf, kb, vb)      # variant 0 = units of 128 rows
    ys = bs_.fprop(x, w); ks = lib.last_kernel()
    dxs = bs_.bprop(e, w)
    assert ks == lib.K_XCOL32_STAGED
    assert torch.equal(y, ys) and torch.equal(dx, dxs)
    # the staged kernel's output against the oracle on sampled columns (the whole tensor is held at N = 1024 above)
    t = orc.build_layout_luts(layout, 32)
    W, X, E = P.to_host(w), P.to_host(x), P.to_host(e)
    dw = P.to_host(b.updat(x, e)); ku = lib.last_kernel()
    assert ku == lib.K_UPDAT_STREAM
    P.assert_blocks(dw, orc.updat_fast(t, X, E, 1, dtype=np.float64), "bf16", b.blocks, ("DW", density))
    bar = P.L2_BAR["bf16"]
    # a few whole output columns of Y / DX against the float64 oracle as well (all 8192 rows)
    for name, got, act, lut_key, nb in (("Y", P.to_host(y), X, "fprop", b.KB), ("DX", P.to_host(dx), E, "bprop", b.CB)):
        full = orc.round_to((orc.fprop_fast if name == "Y" else orc.bprop_fast)(t, act[:512], W, 1, dtype=np.float64), "bf16")
        l2, _ = P.errors(got[:512], full)
        assert l2 <= bar, (name, l2)


# ---- (b) BASELINE configs[3] -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("axis,N,force", [(1, 512, False), (1, 512, True), (1, 4096, False), (0, 512, True), (0, 4096, False)])
def test_cfg3_8192_5pct(env, axis, N, force):
    """8192^2, 5 %: the plan builder picks 16x16 windows (~13 blocks each) on axis 1.  N = 512 is one GPU's shard of the
    4096 minibatch: the cost models may prefer the per-segment / per-block kernels there, so the plan kernels are ALSO run
    forced (BSMM_FLAG_FORCE_PLAN)."""
    torch, BSMM, lib = env
    layout = P.random_layout(256, 256, 0.05, seed=1234)
    b = BSMM(layout, block_size=32, feature_axis=axis)
    expect = {"xprop": _xprop_kernel(lib, axis), "updat": _updat_kernel(lib, axis)} if (force or N >= 4096) else {}
    try:
        lib.set_kernel_variant(3 if force else 0)
        _check_sampled(torch, lib, b, layout, N, "bf16", seed=21, expect=expect, ctx="cfg3 a%d N%d force%d" % (axis, N, force))
    finally:
        lib.set_kernel_variant(0)


# ---- (c) window variants ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("opt", ["PLAN_WINDOW_8", "PLAN_WINDOW_16", "PLAN_WINDOW_16W", "PLAN_STREAM_16", "PLAN_STREAM_8", "PLAN_STREAM_32"])
@pytest.mark.parametrize("density", [0.05, 0.2])
def test_updat_window_variants(env, opt, density):
    torch, BSMM, lib = env
    layout = P.random_layout(128, 128, density, seed=77)
    b = BSMM(layout, block_size=32, feature_axis=1, plan_options=getattr(lib, opt))
    try:
        lib.set_kernel_variant(3)           # the cost model may prefer the per-block kernel for a mismatched window shape
        _check_sampled(torch, lib, b, layout, 2048, "bf16", seed=31, expect={"updat": _updat_kernel(lib, 1, getattr(lib, opt))},
                       ctx="%s d%.2f" % (opt, density), passes=("DW",))
    finally:
        lib.set_kernel_variant(0)


# ---- (d) small forced-plan layouts with several blocks per 16x16 window -----------------------------------------------
@pytest.mark.parametrize("opt", [0, "PLAN_WINDOW_16", "PLAN_WINDOW_16W", "PLAN_STREAM_8", "PLAN_STREAM_32"])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_small_layouts_multi_block_windows(env, opt, dtype):
    torch, BSMM, lib = env
    cases = [(P.random_layout(40, 40, 0.04, seed=1), (72, 200)),        # 64 blocks / 9 windows: default plan = 16x16 windows
             (P.random_layout(40, 40, 0.15, seed=2), (392,)),           # ~27 per window: several items per window when forced
             (P.random_layout(33, 17, 0.3, seed=3), (100, 8)),          # partial windows on both edges
             (P.ba_layout(40, 3, seed=1), (264,))]                      # hub rows: one wave's slots share the X fragment
    try:
        lib.set_kernel_variant(3)
        for li, (layout, Ns) in enumerate(cases):
            b = BSMM(layout, block_size=32, feature_axis=1, plan_options=getattr(lib, opt) if opt else 0)
            t = orc.build_layout_luts(layout, 32)
            for N in Ns:
                W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), dtype, seed=li * 7 + N)
                x, e = P.to_dev(X, dtype, torch), P.to_dev(E, dtype, torch)
                for split in (0, 1, 3):                                 # library's choice / one workgroup per item / three
                    b.updat_split = split
                    got = P.to_host(b.updat(x, e))
                    assert lib.last_kernel() == _updat_kernel(lib, 1, getattr(lib, opt) if opt else 0)
                    l2, mx = P.errors(got, orc.round_to(orc.updat(t, X, E, 1), dtype))
                    assert l2 <= P.L2_BAR[dtype], (opt, li, N, split, l2)
    finally:
        lib.set_kernel_variant(0)


# ---- (c4) small-minibatch weight gradient: one wave per block (round 4) -------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("N", [8, 40, 64, 200, 320, 744])
def test_small_minibatch_updat_one_wave_per_block(env, N, dtype):
    """updat32_a1_small_kernel (bsmm_updat_tr.h): production dispatch for N * pairs <= 768 on feature axis 1 where the cost model
    does not prefer the streaming kernel, bsize 32, 16-bit.  Every block
    against the float64 oracle: plain, alpha / beta accumulate, gated, two pairs while they fit; ragged last chunk (N not a multiple of
    32); a block count that is not a multiple of 4 (the last workgroup has idle waves)."""
    torch, BSMM, lib = env
    for lay in (P.random_layout(40, 24, 0.3, seed=2), P.ba_layout(33, 3, seed=1)):
        b = BSMM(lay, block_size=32, feature_axis=1)
        t = orc.build_layout_luts(np.asarray(lay), 32)
        W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), dtype, seed=N)
        x, e = P.to_dev(X, dtype, torch), P.to_dev(E, dtype, torch)
        got = P.to_host(b.updat(x, e))
        one_wave = (lib.last_kernel(), lib.last_kernel_variant()) == (lib.K_UPDAT_BLOCK_TR, lib.KV_ONE_WAVE)
        assert one_wave or N > 320          # (above a few hundred rows the cost model may prefer the streaming kernel for these small layouts)
        l2, _ = P.errors(got, orc.round_to(orc.updat(t, X, E, 1), dtype))
        assert l2 <= P.L2_BAR[dtype], (N, dtype, l2)
        dw0 = orc.round_to(np.random.RandomState(2).normal(size=b.w_shape).astype(np.float32) * 0.05, dtype)
        gate = np.random.RandomState(3).uniform(0.0, 2.0, size=b.blocks).astype(np.float32)
        got = P.to_host(b.updat(x, e, alpha=0.5, beta=0.25, dw=P.to_dev(dw0, dtype, torch), gate=torch.from_numpy(gate).cuda()))
        assert lib.last_kernel_variant() == lib.KV_ONE_WAVE or N > 320
        l2, _ = P.errors(got, orc.round_to(orc.updat(t, X, E, 1, alpha=0.5, beta=0.25, dw_in=dw0, gate=gate), dtype))
        assert l2 <= P.L2_BAR[dtype], (N, dtype, "alpha/beta/gate", l2)
        if 2 * N <= 768:
            got = P.to_host(b.updat([x, x], [e, e], alpha=0.5))
            assert lib.last_kernel_variant() == lib.KV_ONE_WAVE or 2 * N > 320
            l2, _ = P.errors(got, orc.round_to(orc.updat(t, X, E, 1), dtype))
            assert l2 <= P.L2_BAR[dtype], (N, dtype, "two pairs", l2)


# ---- (c6) fp32 weight gradient at bsize 16 on the windowed kernel (round 4) ------------------------------------------------------------
@pytest.mark.parametrize("axis", [1, 0])
def test_fp32_updat_bsize16_through_the_windowed_kernel(env, axis):
    """bsmm_updat, fp32, bsize 16, feature axis 1 with the windowed 'BSUP' plan (updat16_f32_split: six bf16 piece products as six pairs of one launch, raw
    fp32 sums by the kernel's scratch path, fp32 finalize with alpha / beta / gate): every block against the float64 oracle at the fp32 bar;
    two pairs or a tiny minibatch take the kernels without a plan (same results)."""
    torch, BSMM, lib = env
    for lay, N in ((P.random_layout(80, 80, 0.15, seed=2), 512), (P.random_layout(33, 17, 0.3, seed=3), 264)):
        b = BSMM(lay, block_size=16, feature_axis=axis)
        t = orc.build_layout_luts(np.asarray(lay), 16)
        W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), "f32", seed=31)
        x, e = P.to_dev(X, "f32", torch), P.to_dev(E, "f32", torch)
        got = P.to_host(b.updat(x, e))
        assert (lib.last_kernel() == lib.K_UPDAT16_WIN) == (axis == 1), lib.last_kernel()      # (feature axis 0: the per-block fp32 kernel is as fast)
        l2, _ = P.errors(got, orc.updat(t, X, E, axis))
        assert l2 <= P.L2_BAR["f32"], (axis, l2)
        dw0 = np.random.RandomState(5).normal(size=b.w_shape).astype(np.float32) * 0.05
        gate = np.random.RandomState(6).uniform(0.0, 2.0, size=b.blocks).astype(np.float32)
        got = P.to_host(b.updat(x, e, alpha=0.5, beta=0.25, dw=P.to_dev(dw0, "f32", torch), gate=torch.from_numpy(gate).cuda()))
        assert (lib.last_kernel() == lib.K_UPDAT16_WIN) == (axis == 1)
        l2, _ = P.errors(got, orc.updat(t, X, E, axis, alpha=0.5, beta=0.25, dw_in=dw0, gate=gate))
        assert l2 <= P.L2_BAR["f32"], (axis, "alpha/beta/gate", l2)
        got = P.to_host(b.updat([x, x], [e, e], alpha=0.5))
        assert lib.last_kernel() != lib.K_UPDAT16_WIN
        l2, _ = P.errors(got, orc.updat(t, X, E, axis))
        assert l2 <= P.L2_BAR["f32"], (axis, "two pairs", l2)
        sl = (slice(None), slice(0, 64)) if axis == 0 else (slice(0, 64),)
        got = P.to_host(b.updat(x[sl].contiguous(), e[sl].contiguous()))
        assert lib.last_kernel() != lib.K_UPDAT16_WIN
        l2, _ = P.errors(got, orc.updat(t, X[sl], E[sl], axis))
        assert l2 <= P.L2_BAR["f32"], (axis, "N = 64", l2)


# ---- (c5) fp32 weight gradient at bsize 8 through the bf16 streaming kernel (round 4) -------------------------------------------------
@pytest.mark.parametrize("axis", [1, 0])
def test_fp32_updat_bsize8_through_the_super_block_sums(env, axis):
    """bsmm_updat, fp32, bsize 8 with the 'BSS8' plan (updat8_f32_split: six bf16 piece products as six pairs of one streaming launch over
    the super-blocks, gather8_f32_kernel): every block against the float64 oracle at the fp32 bar, plain and alpha / beta accumulate; two
    pairs, a gate or a tiny minibatch take the V_FMA kernel (same results)."""
    torch, BSMM, lib = env
    for lay, N in ((P.random_layout(64, 64, 0.1, seed=5), 512), (P.ba_layout(32, 3, seed=2), 264)):
        b = BSMM(lay, block_size=8, feature_axis=axis)
        t = orc.build_layout_luts(np.asarray(lay), 8)
        W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), "f32", seed=29)
        x, e = P.to_dev(X, "f32", torch), P.to_dev(E, "f32", torch)
        got = P.to_host(b.updat(x, e))
        assert lib.last_kernel() == lib.K_UPDAT_SUPER8, lib.last_kernel()
        l2, _ = P.errors(got, orc.updat(t, X, E, axis))
        assert l2 <= P.L2_BAR["f32"], (axis, l2)
        dw0 = np.random.RandomState(5).normal(size=b.w_shape).astype(np.float32) * 0.05
        got = P.to_host(b.updat(x, e, alpha=0.5, beta=0.25, dw=P.to_dev(dw0, "f32", torch)))
        assert lib.last_kernel() == lib.K_UPDAT_SUPER8
        l2, _ = P.errors(got, orc.updat(t, X, E, axis, alpha=0.5, beta=0.25, dw_in=dw0))
        assert l2 <= P.L2_BAR["f32"], (axis, "alpha/beta", l2)
        got = P.to_host(b.updat([x, x], [e, e], alpha=0.5))
        assert lib.last_kernel() == lib.K_UPDAT_VALU
        l2, _ = P.errors(got, orc.updat(t, X, E, axis))
        assert l2 <= P.L2_BAR["f32"], (axis, "two pairs", l2)
        sl = (slice(None), slice(0, 64)) if axis == 0 else (slice(0, 64),)
        got = P.to_host(b.updat(x[sl].contiguous(), e[sl].contiguous()))
        assert lib.last_kernel() == lib.K_UPDAT_VALU
        l2, _ = P.errors(got, orc.updat(t, X[sl], E[sl], axis))
        assert l2 <= P.L2_BAR["f32"], (axis, "N = 64", l2)


# ---- (c3) fp32 weight gradient on feature axis 1: six bf16 piece products as six pairs of one streaming launch (round 4) ------------
@pytest.mark.parametrize("case", ["bench_layout", "ragged", "ba", "single"])
def test_fp32_updat_axis1_through_the_streaming_kernel(env, case):
    """bsmm_updat, fp32, bsize 32, feature axis 1 with the streaming plan (updat32_f32_split): every block of DW against the float64 oracle
    at the fp32 bar -- plain, alpha / beta accumulate, gated, and the raw sums + finalize form -- with the streaming kernel asserted; two
    pairs and tiny minibatches take the kernels without a plan (same results)."""
    torch, BSMM, lib = env
    lay, N = {"bench_layout": (P.random_layout(128, 128, 0.2, seed=1234), 1024), "ragged": (P.random_layout(33, 40, 0.3, seed=3), 520),
              "ba": (P.ba_layout(64, 5, seed=1), 384), "single": (np.ones((1, 1), dtype=np.int32), 264)}[case]
    b = BSMM(lay, block_size=32, feature_axis=1)
    t = orc.build_layout_luts(np.asarray(lay), 32)
    W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), "f32", seed=23)
    x, e = P.to_dev(X, "f32", torch), P.to_dev(E, "f32", torch)
    dw = P.to_host(b.updat(x, e))
    assert lib.last_kernel() == lib.K_UPDAT_STREAM
    l2, _ = P.errors(dw, orc.updat(t, X, E, 1))
    assert l2 <= P.L2_BAR["f32"], (case, l2)
    dw0 = np.random.RandomState(5).normal(size=b.w_shape).astype(np.float32) * 0.05
    got = P.to_host(b.updat(x, e, alpha=0.5, beta=0.25, dw=P.to_dev(dw0, "f32", torch)))
    l2, _ = P.errors(got, orc.updat(t, X, E, 1, alpha=0.5, beta=0.25, dw_in=dw0))
    assert l2 <= P.L2_BAR["f32"], (case, "alpha/beta", l2)
    gate = np.random.RandomState(6).uniform(0.0, 2.0, size=b.blocks).astype(np.float32)
    got = P.to_host(b.updat(x, e, gate=torch.from_numpy(gate).cuda()))
    l2, _ = P.errors(got, orc.updat(t, X, E, 1) * gate[:, None, None])
    assert l2 <= P.L2_BAR["f32"], (case, "gated", l2)
    sums = b.updat(x, e, sums_only=True)
    assert sums.dtype == torch.float32 and lib.last_kernel() == lib.K_UPDAT_STREAM
    got = P.to_host(b.updat_finalize(sums, alpha=2.0, dtype=torch.float32))
    l2, _ = P.errors(got, 2.0 * orc.updat(t, X, E, 1))
    assert l2 <= P.L2_BAR["f32"], (case, "sums + finalize", l2)
    # two pairs / a tiny minibatch: the kernels without a plan
    got = P.to_host(b.updat([x, x], [e, e], alpha=0.5))
    assert lib.last_kernel() != lib.K_UPDAT_STREAM
    l2, _ = P.errors(got, orc.updat(t, X, E, 1))
    assert l2 <= P.L2_BAR["f32"], (case, "two pairs", l2)
    got = P.to_host(b.updat(x[:64].contiguous(), e[:64].contiguous()))
    assert lib.last_kernel() != lib.K_UPDAT_STREAM
    l2, _ = P.errors(got, orc.updat(t, X[:64], E[:64], 1))
    assert l2 <= P.L2_BAR["f32"], (case, "N = 64", l2)


# ---- (c2) fp32 on feature axis 1: the activation split fused into the kernel (round 4) ---------------------------------------------
@pytest.mark.parametrize("case", ["odd_in_ragged", "odd_out", "ba", "many_steps", "single", "bench_layout"])
def test_fp32_fused_split_against_the_oracle(env, case):
    """xcol32sf_kernel (bsmm_xcols.h): fp32 slabs staged by LDS-DMA, the three bf16 pieces made between LDS and LDS by the wave that
    requested the rows.  Full-output comparison with the float64 oracle at the fp32 bar for the shapes that stress its edges: an odd
    number of input blocks (the trailing pair has no odd half), ragged minibatch rows (re-read, never stored), an odd number of output
    blocks (partial group), hub columns, a group with more than 64 pair steps (two table batches: the ring is re-primed), one block."""
    torch, BSMM, lib = env
    lay, N = {"odd_in_ragged": (P.random_layout(33, 40, 0.3, seed=3), 520), "odd_out": (P.random_layout(40, 33, 0.3, seed=4), 136),
              "ba": (P.ba_layout(64, 5, seed=1), 256), "many_steps": (P.random_layout(200, 16, 0.1, seed=6), 384),
              "single": (np.ones((1, 1), dtype=np.int32), 40), "bench_layout": (P.random_layout(128, 128, 0.2, seed=1234), 512)}[case]
    b = BSMM(lay, block_size=32, feature_axis=1)
    t = orc.build_layout_luts(np.asarray(lay), 32)
    W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), "f32", seed=17)
    w, x, e = P.to_dev(W, "f32", torch), P.to_dev(X, "f32", torch), P.to_dev(E, "f32", torch)
    try:
        lib.set_kernel_variant(3)
        y = P.to_host(b.fprop(x, w))
        assert lib.last_kernel() == lib.K_XCOL32_F32SPLIT
        dx = P.to_host(b.bprop(e, w))
        assert lib.last_kernel() == lib.K_XCOL32_F32SPLIT
    finally:
        lib.set_kernel_variant(0)
    l2y, _ = P.errors(y, orc.fprop(t, X, W, 1))
    l2x, _ = P.errors(dx, orc.bprop(t, E, W, 1))
    assert l2y <= P.L2_BAR["f32"] and l2x <= P.L2_BAR["f32"], (case, l2y, l2x)


# ---- (d1) streaming updat at the bench layouts with FEW chunks: empty minibatch parts and empty slices ------------------
@pytest.mark.parametrize("density", [0.1, 0.2])
@pytest.mark.parametrize("N", [16, 40, 200])
def test_streaming_updat_partial_sums_with_few_chunks(env, density, N):
    """The kernel writes one region of partial sums per (round, workgroup) and updat2_reduce_kernel walks the same schedule
    to sum them.  With 1 .. 13 chunks of 16 rows for 4 minibatch parts and up to 32 slices per item of the last round, most
    (part, slice) ranges are EMPTY -- both sides must skip exactly the same ones.  Also two (x, dy) pairs and beta != 0."""
    torch, BSMM, lib = env
    layout = P.random_layout(128, 128, density, seed=1234)
    b = BSMM(layout, block_size=32, feature_axis=1)
    try:
        lib.set_kernel_variant(3)
        _check_sampled(torch, lib, b, layout, N, "bf16", seed=31 + N, expect={"updat": lib.K_UPDAT_STREAM}, ctx="few chunks d%d N%d" % (round(density * 100), N),
                       passes=("DW",))
        t = orc.build_layout_luts(layout, 32)
        Xs, Es = [], []
        for p in range(2):
            _, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), "bf16", seed=50 + p)
            Xs.append(X); Es.append(E)
        dw0 = orc.round_bf16(np.random.RandomState(2).normal(size=b.w_shape).astype(np.float32) * 0.05)
        out = P.to_host(b.updat([P.to_dev(x, "bf16", torch) for x in Xs], [P.to_dev(e, "bf16", torch) for e in Es], alpha=0.25, beta=1.5,
                                dw=P.to_dev(dw0, "bf16", torch)))
        assert lib.last_kernel() == lib.K_UPDAT_STREAM
        l2, _ = P.errors(out, orc.round_bf16(orc.updat(t, Xs, Es, 1, alpha=0.25, beta=1.5, dw_in=dw0)))
        assert l2 <= P.L2_BAR["bf16"], l2
    finally:
        lib.set_kernel_variant(0)


# ---- (d2) xprop plan kernels on small forced layouts: staged (default) and round-1 ------------------------------------
@pytest.mark.parametrize("axis", [1, 0])
@pytest.mark.parametrize("opt", [0, "PLAN_XCOL_UNSTAGED", "PLAN_XCOL_NARROW"])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_small_layouts_xprop_plan_kernels(env, opt, dtype, axis):
    """fprop (in-kernel transposing reads of the staged weight blocks) and bprop against the full float64 oracle: dense layouts
    (steps with 32 blocks are split over phases), odd block counts (half-empty last pair), partial last group, one block,
    minibatches that are not a multiple of the 128-row tile (axis 0: multiples of 8, the plan kernels' alignment rule)."""
    torch, BSMM, lib = env
    o = getattr(lib, opt) if opt else 0
    cases = [(np.ones((9, 35), dtype=bool), (72, 200)),                 # dense: 32 blocks per step > 24 slots per ring half
             (P.random_layout(33, 17, 0.3, seed=3), (104, 8)),          # odd block counts on both sides
             (np.ones((1, 1), dtype=bool), (40,)),                      # one block, one group of one column
             (P.random_layout(40, 40, 0.15, seed=2), (392, 128)),
             (P.ba_layout(40, 3, seed=1), (264,))]
    try:
        lib.set_kernel_variant(3)
        for li, (layout, Ns) in enumerate(cases):
            b = BSMM(layout, block_size=32, feature_axis=axis, plan_options=o)
            t = orc.build_layout_luts(layout, 32)
            for N in Ns:
                W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), dtype, seed=li * 5 + N)
                w, x, e = P.to_dev(W, dtype, torch), P.to_dev(X, dtype, torch), P.to_dev(E, dtype, torch)
                y = P.to_host(b.fprop(x, w))
                assert lib.last_kernel() == _xprop_kernel(lib, axis, o)
                dx = P.to_host(b.bprop(e, w))
                assert lib.last_kernel() == _xprop_kernel(lib, axis, o)
                l2y, _ = P.errors(y, orc.round_to(orc.fprop(t, X, W, axis), dtype))
                l2x, _ = P.errors(dx, orc.round_to(orc.bprop(t, E, W, axis), dtype))
                assert l2y <= P.L2_BAR[dtype] and l2x <= P.L2_BAR[dtype], (opt, axis, li, N, l2y, l2x)
    finally:
        lib.set_kernel_variant(0)


@pytest.mark.parametrize("axis", [1, 0])
def test_staged_and_round1_xprop_kernels_agree_bitwise(env, axis):
    """Both plan kernels accumulate a block column in the same order (pairs ascending, even half first) with the same MFMAs:
    identical bits at the bench shape, fprop and bprop, either feature axis."""
    torch, BSMM, lib = env
    layout = P.random_layout(128, 128, 0.2, seed=1234)
    outs = []
    for o, flow in ((0, True), (0, False), (lib.PLAN_XCOL_UNSTAGED, False)):      # flow (axis 1) | staged | round-1 kernel
        b = BSMM(layout, block_size=32, feature_axis=axis, plan_options=o)
        b.flow = flow
        w, x, e = _inputs(torch, b, 8192, "bf16", seed=5)
        y = b.fprop(x, w); k1 = lib.last_kernel()
        dx = b.bprop(e, w)
        want = _xprop_kernel(lib, axis, o) if flow or o else lib.K_XCOL32_STAGED
        assert k1 == lib.last_kernel() == want, (k1, want)
        outs.append((y, dx))
    for y, dx in outs[1:]:
        assert torch.equal(outs[0][0], y) and torch.equal(outs[0][1], dx)


@pytest.mark.parametrize("axis", [1, 0])
@pytest.mark.parametrize("opt", [0, "PLAN_XCOL_UNSTAGED"])
def test_small_layouts_bsize16_xprop_plan_kernels(env, opt, axis):
    """bsize 16: the staged / list kernels ('BSX7' plans: weight blocks by LDS-DMA, two per instruction) against the full oracle -- dense
    layouts (steps of 128 blocks are split), block counts that are not multiples of 4 (trailing quad with missing blocks), partial last
    group, ragged minibatch.  BSMM_PLAN_XCOL_UNSTAGED named the round-1 kernel (retired in round 4) and is ignored: the same kernels."""
    torch, BSMM, lib = env
    o = getattr(lib, opt) if opt else 0
    want_k = lib.K_XCOL16_STAGED
    cases = [(np.ones((9, 70), dtype=bool), (72, 200)), (P.random_layout(33, 17, 0.3, seed=3), (104, 8)), (np.ones((1, 1), dtype=bool), (40,)),
             (P.random_layout(80, 80, 0.15, seed=2), (392, 128)), (P.ba_layout(80, 3, seed=1), (264,))]
    try:
        lib.set_kernel_variant(3)
        for li, (layout, Ns) in enumerate(cases):
            b = BSMM(layout, block_size=16, feature_axis=axis, plan_options=o)
            t = orc.build_layout_luts(layout, 16)
            for N in Ns:
                for dtype in ("bf16", "f16"):
                    W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), dtype, seed=li * 3 + N)
                    w, x, e = P.to_dev(W, dtype, torch), P.to_dev(X, dtype, torch), P.to_dev(E, dtype, torch)
                    y = P.to_host(b.fprop(x, w))
                    assert lib.last_kernel() == want_k
                    dx = P.to_host(b.bprop(e, w))
                    assert lib.last_kernel() == want_k
                    l2y, _ = P.errors(y, orc.round_to(orc.fprop(t, X, W, axis), dtype))
                    l2x, _ = P.errors(dx, orc.round_to(orc.bprop(t, E, W, axis), dtype))
                    assert l2y <= P.L2_BAR[dtype] and l2x <= P.L2_BAR[dtype], (opt, axis, li, N, dtype, l2y, l2x)
    finally:
        lib.set_kernel_variant(0)


def test_bsize64_axis1(env):
    """bsize 64 on feature_axis 1 (the reference's second axis-1 block size, blocksparse/matmul.py:84-89): the host class runs it on the
    bsize-32 kernels (a 64x64 block = four 32x32 blocks) -- fprop / bprop / updat with alpha, beta and a gate against the oracle at bsize 64."""
    torch, BSMM, lib = env
    layout = P.random_layout(24, 40, 0.25, seed=9)
    b = BSMM(layout, block_size=64, feature_axis=1)
    assert b.w_shape == (int(layout.sum()), 64, 64) and b.C == 24 * 64 and b.K == 40 * 64
    t = orc.build_layout_luts(layout, 64)
    for dtype in ("bf16", "f32"):
        for N in (72, 520):
            W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), dtype, seed=3 + N)
            w, x, e = P.to_dev(W, dtype, torch), P.to_dev(X, dtype, torch), P.to_dev(E, dtype, torch)
            g = np.random.default_rng(N).random(b.blocks).astype(np.float32)
            g[::3] = 0
            tg = torch.from_numpy(g).cuda()
            dw0 = orc.round_to(np.random.default_rng(1).normal(size=b.w_shape).astype(np.float32) * 0.05, dtype)
            got = {"Y": b.fprop(x, w), "DX": b.bprop(e, w), "Yg": b.fprop(x, w, gate=tg), "DW": b.updat(x, e),
                   "DWab": b.updat(x, e, alpha=0.5, beta=2.0, dw=P.to_dev(dw0, dtype, torch))}
            want = {"Y": orc.fprop(t, X, W, 1), "DX": orc.bprop(t, E, W, 1), "Yg": orc.fprop(t, X, W, 1, gate=g), "DW": orc.updat(t, X, E, 1),
                    "DWab": orc.updat(t, X, E, 1, alpha=0.5, beta=2.0, dw_in=dw0)}
            for k in got:
                l2, _ = P.errors(P.to_host(got[k]), orc.round_to(want[k], dtype))
                assert l2 <= P.L2_BAR[dtype], (dtype, N, k, l2)
    with pytest.raises(ValueError):
        BSMM(layout, block_size=64, feature_axis=0)
    # round 3: the calls above went to the library with bsmm_args.bsize = 64 ('BS64' plans; fp32 updat excepted).  The host-side
    # quadrant view (native64 = False) is the same arithmetic on the same kernels: bit-identical for xprop, and for the 16-bit updat
    # up to the order in which the streaming kernel's partial sums meet
    W, X, E = P.make_inputs(b.w_shape, b.i_shape(520), b.o_shape(520), "bf16", seed=77)
    w, x, e = P.to_dev(W, "bf16", torch), P.to_dev(X, "bf16", torch), P.to_dev(E, "bf16", torch)
    y1, dx1, dw1 = b.fprop(x, w), b.bprop(e, w), b.updat(x, e)
    assert lib.last_kernel() == lib.K_UPDAT_STREAM
    b.native64 = False
    y0, dx0, dw0 = b.fprop(x, w), b.bprop(e, w), b.updat(x, e)
    b.native64 = True
    assert torch.equal(y1, y0) and torch.equal(dx1, dx0)
    assert ((dw1.float() - dw0.float()).norm() / dw0.float().norm()).item() < 1e-3


def test_bsize64_axis1_helper_ops(env):
    """bsize 64: the ops around the matmul run on the quadrant view too -- gate gradient (and with it the gated autograd
    path), identity init, L2 weight norm, raw fp32 sums + finalize; each against the same op at bsize 32 on the four-quadrant layout or
    against NumPy.  The quadrant view of a constant W is made once per tensor version."""
    torch, BSMM, lib = env
    layout = P.random_layout(12, 12, 0.3, seed=19)
    layout[np.arange(12), np.arange(12)] = 1
    b = BSMM(layout, block_size=64, feature_axis=1)
    rng = np.random.default_rng(5)
    W = torch.from_numpy(rng.normal(size=b.w_shape).astype(np.float32) * 0.05).cuda().bfloat16()
    DW = torch.from_numpy(rng.normal(size=b.w_shape).astype(np.float32) * 0.05).cuda().bfloat16()
    g = torch.from_numpy(rng.random(b.blocks).astype(np.float32)).cuda()
    out, dg = b.gate_grad(DW, W, g)
    want_dg = (DW.float() * W.float()).sum(dim=(1, 2))
    assert torch.allclose(dg, want_dg, rtol=1e-4, atol=1e-5)
    assert torch.equal(out, (DW.float() * g[:, None, None]).to(torch.bfloat16))
    # identity init
    I = b.identity_init(0.5)(dtype=torch.float32)
    eye = torch.zeros(b.w_shape)
    for w_, (c, k) in enumerate(b.updat_list):
        if c == k:
            eye[w_] = 0.5 * torch.eye(64)
    assert torch.equal(I.cpu(), eye)
    # L2 norm against the NumPy restatement of the class
    Wf = W.float()
    y = b.l2_normalize(Wf)
    assert np.allclose(y.cpu().numpy(), b.l2_normalize_test(Wf.cpu().numpy()), rtol=1e-5, atol=1e-6)
    # sums + finalize == updat (one rounding each)
    x = (torch.randn(b.i_shape(256), device="cuda", generator=P.gen(torch, 71)) * 0.1).bfloat16()
    dy = (torch.randn(b.o_shape(256), device="cuda", generator=P.gen(torch, 72)) * 0.1).bfloat16()
    sums = b.updat(x, dy, sums_only=True)
    assert sums.dtype == torch.float32 and tuple(sums.shape) == b.w_shape
    fin = b.updat_finalize(sums, alpha=0.5, beta=2.0, dw=DW.clone(), gate=g)
    ref = b.updat(x, dy, alpha=0.5, beta=2.0, dw=DW.clone(), gate=g)
    d = (fin.float() - ref.float()).abs()
    assert (d.norm() / ref.float().norm()).item() < 1e-3
    # gated autograd path
    xg = x.float().requires_grad_(True); wg = W.float().requires_grad_(True); gg = g.clone().requires_grad_(True)
    yv = b(xg, wg, gate=gg, gate_grad=True)
    yv.backward(torch.ones_like(yv))
    assert xg.grad is not None and wg.grad is not None and gg.grad is not None and torch.isfinite(gg.grad).all()
    # the quadrant copy of a constant W is made once per (op, storage, version): by the library call path (bsmm_prepare_weights) ...
    b.fprop(x, W); key = b._prepared_w[lib.OP_FPROP][1]
    b.fprop(x, W); assert b._prepared_w[lib.OP_FPROP][1] == key
    W.add_(0.0); b.fprop(x, W); assert b._prepared_w[lib.OP_FPROP][1] != key
    # ... and by the host-side quadrant view
    b.native64 = False
    b.fprop(x, W); first = b._split64_hit[2]
    b.bprop(dy, W); assert b._split64_hit[2] is first
    W.add_(0.0); b.fprop(x, W); assert b._split64_hit[2] is not first
    # a NEW tensor at a recycled address (same data_ptr, same version) must not hit either cache
    for native in (True, False):
        b.native64 = native
        b.invalidate_weights()
        W1 = torch.randn(W.shape, device=W.device, generator=P.gen(torch, 73)).to(W.dtype) * 0.05
        y_1 = b.fprop(x, W1)
        del W1
        W2 = torch.empty_like(W)                          # the caching allocator hands the freed block back
        W2.copy_(torch.randn(W.shape, device=W.device, generator=P.gen(torch, 74)).to(W.dtype) * 0.05)
        y_2 = b.fprop(x, W2)
        assert torch.equal(y_2, b.fprop(x, W2.clone())), native
        assert not torch.equal(y_1, y_2)
    b.native64 = True


# ---- (e) the reference's own test matrix ------------------------------------------------------------------------------
@pytest.mark.parametrize("axis", [0, 1])
@pytest.mark.parametrize("bs", [32, 16, 8])
def test_reference_matrix_ba160(env, bs, axis):
    """BA(n=160, m=5) + I with a dense 5x5 corner, N in {256,128,64,32,16,8}, fp32 (the reference's dtype for this test) and
    bf16 at the two ends; the whole output against the oracle (test/blocksparse_matmul_test.py:276-328)."""
    torch, BSMM, lib = env
    layout = P.ba_layout(160, 5, seed=1)
    for N in (256, 128, 64, 32, 16, 8):
        for dtype in ("f32",) + (("bf16",) if N in (256, 8) else ()):
            res = P.run_case(torch, BSMM, layout, bs, axis, dtype, N, seed=N + bs, fast_oracle=True)
            for name, (l2, mx) in res.items():
                assert l2 <= P.L2_BAR[dtype], ("ba160", bs, axis, dtype, N, name, l2)
    # and through the reference-policy (segmented, locked) tables, as a reference-side binding without a plan would pass them
    b = BSMM(layout, block_size=bs, feature_axis=axis, segmented=True)
    assert b.fprop_locks > 0
    res = P.run_case(torch, BSMM, layout, bs, axis, "bf16", 64, seed=5, segmented=True, passes=("Y", "DX"), fast_oracle=True)
    for name, (l2, mx) in res.items():
        assert l2 <= P.L2_BAR["bf16"], ("ba160 locked", bs, axis, name, l2)


# ---- (f) BASELINE configs[2] ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("axis", [0, 1])
def test_cfg2_bsize16_10pct(env, axis):
    torch, BSMM, lib = env
    layout = P.random_layout(256, 256, 0.1, seed=1234)
    b = BSMM(layout, block_size=16, feature_axis=axis)
    _check_sampled(torch, lib, b, layout, 8192, "bf16", seed=41, expect={"xprop": lib.K_XCOL16_STAGED, "updat": lib.K_UPDAT16_ROWS if axis == 0 else lib.K_UPDAT16_WIN}, ctx="cfg2 a%d" % axis)      # (feature axis 0: the row-owner kernel, round 5)


def test_cfg1_fp32_axis1(env):
    """BASELINE configs[1]: 4096^2, bsize 32, 20 %, fp32, feature axis 1, fprop (the exact bf16-split kernel), N = 8192."""
    torch, BSMM, lib = env
    layout = P.random_layout(128, 128, 0.2, seed=1234)
    b = BSMM(layout, block_size=32, feature_axis=1)
    _check_sampled(torch, lib, b, layout, 8192, "f32", seed=51, expect={"xprop": lib.K_XCOL32_F32SPLIT}, ctx="cfg1", passes=("Y", "DX"))


# ---- (g) bsize 8 at the BASELINE scale (north_star names bsize 8 on both axes) -----------------------------------------------
@pytest.mark.parametrize("axis", [0, 1])
def test_bench_shape_bsize8(env, axis):
    """4096^2, bsize 8 (512 x 512 blocks), 10 %, bf16, N = 8192, through the 'BSS8' super-block path (the 8x8 blocks grouped into
    32x32 super-blocks on the bsize-32 matrix-core kernels): sampled output block columns / rows / weight blocks vs the float64
    oracle (blocksparse/matmul.py:353-419), the kernel family asserted."""
    torch, BSMM, lib = env
    layout = P.random_layout(512, 512, 0.1, seed=1234)
    b = BSMM(layout, block_size=8, feature_axis=axis)
    _check_sampled(torch, lib, b, layout, 8192, "bf16", seed=71, expect={"xprop": lib.K_XPROP_SUPER8, "updat": lib.K_UPDAT_SUPER8},
                   ctx="bs8 a%d" % axis, n_cols=16, n_blocks=128)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_small_minibatch_kernel(env, dtype):
    """bsmm_xsmall.h (round 4): feature axis 1, bsize 32, production dispatch at small minibatches -- every output element of fprop and
    bprop against the float64 oracle, the kernel asserted; layouts with empty rows / columns, a single block, more entries per column
    than waves, fewer entries than waves; ragged minibatches (not a multiple of the 64-row tile, smaller than a tile)."""
    torch, BSMM, lib = env
    lone = np.zeros((9, 11), dtype=np.int32); lone[3, 7] = 1
    cases = [(P.ba_layout(160, 5, seed=0), (64, 37, 200)), (P.random_layout(40, 24, 0.5, seed=3), (8, 130)), (lone, (64,)),
             (np.ones((20, 3), dtype=np.int32), (100,)), (P.random_layout(128, 128, 0.2, seed=1234), (128,))]
    for li, (lay, Ns) in enumerate(cases):
        b = BSMM(lay, block_size=32, feature_axis=1)
        t = orc.build_layout_luts(lay, 32)
        for N in Ns:
            W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), dtype, seed=li * 7 + N)
            w, x, e = P.to_dev(W, dtype, torch), P.to_dev(X, dtype, torch), P.to_dev(E, dtype, torch)
            y = P.to_host(b.fprop(x, w)); kf = lib.last_kernel()
            dx = P.to_host(b.bprop(e, w)); kb = lib.last_kernel()
            assert kf == kb == lib.K_XPROP_SMALL, (li, N, kf, kb)
            l2y, _ = P.errors(y, orc.round_to(orc.fprop(t, X, W, 1), dtype))
            l2x, _ = P.errors(dx, orc.round_to(orc.bprop(t, E, W, 1), dtype))
            assert l2y <= P.L2_BAR[dtype] and l2x <= P.L2_BAR[dtype], (li, N, l2y, l2x)
            # output blocks without entries are zero, not garbage
            if li == 2:
                assert np.count_nonzero(y[:, :7 * 32]) == 0 and np.count_nonzero(y[:, 8 * 32:]) == 0


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_medium_minibatch_kernel(env, dtype):
    """bsmm_xmid.h (round 4): feature axis 1, bsize 32 -- one wave per (output block, 64 rows) walks the block's lut segment from a
    private LDS ring.  Forced per call (BSMM_FLAG_FORCE_MID): every output element of fprop and bprop against the float64 oracle AND
    bit for bit against the plan kernels (same summation order: the lut's), both workgroup shapes (one block x 4 row chunks for small
    activation matrices, 4 blocks x one row chunk above 5 MB); layouts with empty columns, a single block, more than 64 entries per
    column (two chunks of the lane-resident list), fewer entries than ring stages; ragged minibatches."""
    torch, BSMM, lib = env
    lone = np.zeros((9, 11), dtype=np.int32); lone[3, 7] = 1
    cases = [(P.ba_layout(160, 5, seed=0), (64, 37, 200)), (P.random_layout(40, 24, 0.5, seed=3), (8, 130, 1000)), (lone, (64,)),
             (np.ones((20, 3), dtype=np.int32), (100,)), (P.random_layout(128, 128, 0.2, seed=1234), (512,)),
             (P.random_layout(128, 128, 0.55, seed=5), (300,)), (P.random_layout(256, 256, 0.05, seed=1234), (520,)), (np.eye(15, 33, dtype=np.int32), (70,))]
    for li, (lay, Ns) in enumerate(cases):
        b = BSMM(lay, block_size=32, feature_axis=1)
        t = orc.build_layout_luts(lay, 32)
        for N in Ns:
            W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), dtype, seed=li * 7 + N)
            w, x, e = P.to_dev(W, dtype, torch), P.to_dev(X, dtype, torch), P.to_dev(E, dtype, torch)
            lib.set_kernel_variant(4)
            try:
                ty = b.fprop(x, w); kf = lib.last_kernel()
                tdx = b.bprop(e, w); kb = lib.last_kernel()
                lib.set_kernel_variant(3)
                py, pdx = b.fprop(x, w), b.bprop(e, w)
            finally:
                lib.set_kernel_variant(0)
            assert kf == kb == lib.K_XPROP_MID, (li, N, kf, kb)
            assert torch.equal(ty, py) and torch.equal(tdx, pdx), (li, N)
            y, dx = P.to_host(ty), P.to_host(tdx)
            l2y, _ = P.errors(y, orc.round_to(orc.fprop(t, X, W, 1), dtype))
            l2x, _ = P.errors(dx, orc.round_to(orc.bprop(t, E, W, 1), dtype))
            assert l2y <= P.L2_BAR[dtype] and l2x <= P.L2_BAR[dtype], (li, N, l2y, l2x)
            if li == 2:        # output blocks without entries are zero, not garbage
                assert np.count_nonzero(y[:, :7 * 32]) == 0 and np.count_nonzero(y[:, 8 * 32:]) == 0


# ---- (h) the kernel-choice cost models at measured points (profiles/r02_sweeps.md: the faster kernel wins by > 15 % there) ----
@pytest.mark.parametrize("CB,dens,N,xk,uk", [(128, 0.2, 256, None, "K_UPDAT_BLOCK_TR"), (128, 0.2, 2048, "K_XCOL32_FLOW", "K_UPDAT_STREAM"),
                                             (128, 0.05, 2048, None, "K_UPDAT_STREAM"), (128, 0.05, 512, None, "K_UPDAT_BLOCK_TR"),
                                             (256, 0.05, 512, "K_XPROP_MID", "K_UPDAT_STREAM"), (256, 0.05, 2048, "K_XCOL32_FLOW", "K_UPDAT_STREAM"),
                                             (64, 0.2, 512, None, "K_UPDAT_BLOCK_TR"), (64, 0.2, 8192, "K_XCOL32_FLOW", None),
                                             (128, 0.2, 64, "K_XPROP_SMALL", None), (128, 0.2, 512, "K_XPROP_MID", None), (128, 0.2, 4096, "K_XCOL32_FLOW", None)])
def test_cost_models_pick_the_measured_winner(env, CB, dens, N, xk, uk):
    """Production dispatch (no flags) on a 256-CU part: at these (layout, minibatch) points of the sweeps (round 4: timed as hipGraph
    replays, profiles/r04_smalln.txt) one kernel family is clearly faster; (128, 5 %, N = 2048) is the updat point the model once got wrong (auto 33.6 us on the per-block kernel, plan 25.2)."""
    torch, BSMM, lib = env
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("the measured points are those of a 256-CU part")
    layout = P.random_layout(CB, CB, dens, seed=1234)
    b = BSMM(layout, block_size=32, feature_axis=1)
    w, x, e = _inputs(torch, b, N, "bf16", 3)
    b.bprop(e, w)
    if xk is not None:
        assert lib.last_kernel() == getattr(lib, xk), ("bprop", CB, dens, N, lib.last_kernel())
    b.updat(x, e)
    if uk is not None:
        assert lib.last_kernel() == getattr(lib, uk), ("updat", CB, dens, N, lib.last_kernel())
    torch.cuda.synchronize()


# ---- boundary: a plan that does not belong to the call is refused, not silently ignored -------------------------------
def test_mismatched_plan_is_rejected(env):
    import ctypes
    torch, BSMM, lib = env
    layout = P.random_layout(16, 16, 0.3, seed=4)
    b = BSMM(layout, block_size=32, feature_axis=1)
    N = 256
    w, x, e = _inputs(torch, b, N, "bf16", 1)
    tabs = b._tables_on(x.device)
    L = lib.load()
    y = torch.full(b.o_shape(N), 7.0, dtype=torch.bfloat16, device="cuda")

    def args(plan, lut, side):
        a = b._args(lut, side, N, b.K, b.C, torch.bfloat16, plan=plan)
        a.flags = lib.FLAG_FORCE_PLAN
        return a
    # an updat plan handed to bprop
    a = args(tabs.updat_plan, tabs.bprop, b._dev_tables["bprop"])
    assert L.bsmm_bprop(e.data_ptr(), w.data_ptr(), y.data_ptr(), ctypes.byref(a)) == -1
    # the right format with a descriptor of another width (a plan built with other options than the caller claims)
    a = args(tabs.bprop_plan, tabs.bprop, b._dev_tables["bprop"])
    a.plan_width = 5
    assert L.bsmm_bprop(e.data_ptr(), w.data_ptr(), y.data_ptr(), ctypes.byref(a)) == -1
    # an xprop plan handed to updat
    a = b._args(tabs.updat, None, N, b.C, b.K, torch.bfloat16, plan=tabs.fprop_plan)
    a.flags = lib.FLAG_FORCE_PLAN
    dw = torch.empty(b.w_shape, dtype=torch.bfloat16, device="cuda")
    arr = (ctypes.c_void_p * 1)
    assert L.bsmm_updat(arr(x.data_ptr()), arr(e.data_ptr()), dw.data_ptr(), ctypes.byref(a)) == -1
    torch.cuda.synchronize()
    assert (y.float() == 7.0).all()            # the refused calls launched nothing


def test_locked_tables_fp32_axis0_unaligned_minibatch(env):
    """Regression (round-1 advisor finding): fp32, bsize 32, feature axis 0, reference-policy (locked) tables, a minibatch
    with N % 8 == 4 large enough for the grouped-kernel heuristic.  The split kernel needs N % 8 == 0 on axis 0, so the call
    must take the per-segment kernel AND zero-fill the locked output blocks first (the two decisions used to disagree)."""
    torch, BSMM, lib = env
    layout = P.ba_layout(128, 6, seed=3)
    b = BSMM(layout, block_size=32, feature_axis=0, segmented=True)
    assert b.fprop_locks > 0 and b.bprop_locks > 0
    for N in (2052, 2056):
        expect = {"xprop": lib.K_XPROP_SEGMENT if N % 8 else lib.K_XCOL32_F32SPLIT}
        _check_sampled(torch, lib, b, layout, N, "f32", seed=61, expect=expect, ctx="locked f32 a0 N%d" % N, passes=("Y", "DX"))
