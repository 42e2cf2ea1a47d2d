"""Per-block gating: CPU tier pins oracle + host NumPy methods to fixtures generated from the
reference (tests/golden/make_golden.py gate); GPU tier checks the gated kernels, the gate gradient and the autograd wiring."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as MG
import _parity as P
from oracle import bsmm_oracle as O


def _case(gold, bs):
    key = "holes/bs%d/" % bs
    lay = gold[key + "layout"].astype(np.int32)
    _, N, seed = (int(v) for v in gold[key + "meta"])
    t = O.build_layout_luts(lay, bs)
    W, X, E = MG.cfg0_inputs((t["blocks"], bs, bs), (t["C"], N), (t["K"], N), seed)
    return key, lay, t, W, X, E, MG.gate_inputs(t["blocks"], seed)


@pytest.mark.parametrize("bs", [32, 16, 8])
def test_oracle_and_host_methods_match_reference_fixtures(bs):
    from blocksparse_amd import BlocksparseMatMul
    gold = np.load(os.path.join(HERE, "golden", "gate.npz"))
    key, lay, t, W, X, E, g = _case(gold, bs)
    b = BlocksparseMatMul(lay, block_size=bs, feature_axis=0)
    assert (g == 0).sum() == t["blocks"] // 3
    for name, ref, host, orc in (("Y", gold[key + "Y"], b.fprop_test(X, W, gate=g), O.fprop(t, X, W, 0, gate=g)),
                                 ("DX", gold[key + "DX"], b.bprop_test(E, W, gate=g), O.bprop(t, E, W, 0, gate=g)),
                                 ("DW", gold[key + "DW"], b.updat_test(X, E, gate=g, dw_gated=True), O.updat(t, X, E, 0, gate=g))):
        for what, got in (("host", host), ("oracle", orc)):
            err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
            assert err < 1e-6, (bs, name, what, err)
    # blocks with gate 0 contribute nothing: same result with those weights replaced by garbage
    W2 = W.copy()
    W2[g == 0] = 1e6
    assert np.array_equal(O.fprop(t, X, W, 0, gate=g), O.fprop(t, X, W2, 0, gate=g))


def test_prune_drops_gated_off_blocks():
    from blocksparse_amd import BlocksparseMatMul
    lay = P.ba_layout(12, 2, seed=4)
    b = BlocksparseMatMul(lay, block_size=8, feature_axis=0)
    rng = np.random.RandomState(0)
    W = rng.normal(size=b.w_shape).astype(np.float32)
    g = np.ones(b.blocks, dtype=np.float32)
    off = rng.permutation(b.blocks)[:5]
    g[off] = 0
    W2, g2 = b.prune(W, g)
    assert W2.shape[0] == b.blocks - 5 and g2.shape == (b.blocks - 5,) and np.all(g2 == 1)
    b2 = BlocksparseMatMul(b.layout, block_size=8, feature_axis=0)        # rebuilt from the pruned layout
    assert b2.blocks == b.blocks - 5
    x = rng.normal(size=b.i_shape(16)).astype(np.float32)
    assert np.allclose(b2.fprop_test(x, W2), b.fprop_test(x, W, gate=g), atol=1e-5)


# ---------------------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def env():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import __graft_entry__ as g
    g.build()
    from blocksparse_amd import BlocksparseMatMul
    return torch, BlocksparseMatMul


def _t(torch, a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda().to(getattr(torch, P.TORCH_DT[dtype]))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("axis", [0, 1])
@pytest.mark.parametrize("bs", [32, 16, 8])
def test_gated_passes_match_oracle(env, bs, axis, dtype):
    torch, BSMM = env
    holes = P.ba_layout(16, 2, seed=3)
    holes[:, 5] = 0
    holes[7, :] = 0
    for li, lay in enumerate((holes, P.random_layout(9, 7, 0.4, seed=8))):
        for N in (8, 72, 200):
            b = BSMM(lay, block_size=bs, feature_axis=axis)
            t = O.build_layout_luts(lay, bs)
            W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), dtype, seed=40 + N)
            g = MG.gate_inputs(b.blocks, 50 + li)
            tg = torch.from_numpy(g).cuda()
            tw, tx, te = _t(torch, W, dtype), _t(torch, X, dtype), _t(torch, E, dtype)
            rnd = lambda a: O.round_to(a, dtype)
            for name, got, ref in (("Y", b.fprop(tx, tw, gate=tg), rnd(O.fprop(t, X, W, axis, gate=g))),
                                   ("DX", b.bprop(te, tw, gate=tg), rnd(O.bprop(t, E, W, axis, gate=g))),
                                   ("DW", b.updat(tx, te, gate=tg), rnd(O.updat(t, X, E, axis, gate=g)))):
                l2, _ = P.errors(got.float().cpu().numpy(), ref)
                assert l2 <= P.L2_BAR[dtype], (bs, axis, dtype, li, N, name, l2)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("axis", [1, 0])
def test_gated_bsize16(env, axis, dtype):
    """Gated bsize-16 calls (the reference gates all of its tensor-core block sizes, src/blocksparse_hgemm_cn_64_op_gpu.cu:256-717).  Round 6: the
    GATED instantiation of the pair kernel is gone (22 spilled registers, 3.3x the ungated time); with a plan forced on small layouts -- gates 0, 1,
    negative, > 1 -- a gated call runs the per-segment kernel, and at BASELINE configs[2]'s shape the list kernel over gated weight images (the
    default) as well as the per-segment kernel (gate_images = False), all against the oracle, which gates the fp32 block product."""
    torch, BSMM = env
    from blocksparse_amd import _lib
    cases = [(P.random_layout(18, 14, 0.4, seed=8), (72, 200)), (np.ones((6, 40), dtype=bool), (128,)), (P.ba_layout(60, 3, seed=1), (264,))]
    try:
        _lib.set_kernel_variant(3)
        for li, (lay, Ns) in enumerate(cases):
            b = BSMM(lay, block_size=16, feature_axis=axis)
            t = O.build_layout_luts(lay, 16)
            for N in Ns:
                W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), dtype, seed=60 + N)
                g = MG.gate_inputs(b.blocks, 70 + li)
                tg = torch.from_numpy(g).cuda()
                tw, tx, te = _t(torch, W, dtype), _t(torch, X, dtype), _t(torch, E, dtype)
                y = b.fprop(tx, tw, gate=tg)
                assert _lib.last_kernel() == _lib.K_XPROP_SEGMENT
                dx = b.bprop(te, tw, gate=tg)
                assert _lib.last_kernel() == _lib.K_XPROP_SEGMENT
                for name, got, ref in (("Y", y, O.fprop(t, X, W, axis, gate=g)), ("DX", dx, O.bprop(t, E, W, axis, gate=g))):
                    l2, _ = P.errors(got.float().cpu().numpy(), O.round_to(ref, dtype))
                    assert l2 <= P.L2_BAR[dtype], (axis, dtype, li, N, name, l2)
    finally:
        _lib.set_kernel_variant(0)
    lay = P.random_layout(256, 256, 0.1, seed=1234)
    b = BSMM(lay, block_size=16, feature_axis=axis)
    t = O.build_layout_luts(lay, 16)
    N = 4096
    W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), dtype, seed=7)
    g = MG.gate_inputs(b.blocks, 9)
    for images, family in ((True, _lib.K_XCOL16_STAGED), (False, _lib.K_XPROP_SEGMENT)):
        b.gate_images = images
        y = b.fprop(_t(torch, X, dtype), _t(torch, W, dtype), gate=torch.from_numpy(g).cuda())
        assert _lib.last_kernel() == family
        yh = y.float().cpu().numpy()
        for k, ref in O.fprop_cols(t, X * 1.0, np.asarray(W, dtype=np.float64) * g[:, None, None], axis, [0, 77, 255]).items():
            got = yh[:, k * 16:(k + 1) * 16] if axis else yh[k * 16:(k + 1) * 16, :]
            l2, _ = P.errors(got, O.round_to(ref, dtype))
            assert l2 <= P.L2_BAR[dtype], (axis, dtype, images, k, l2)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("axis", [1, 0])
def test_gated_staged_plan_kernel(env, axis, dtype):
    """Gates on the bsize-32 plan kernel (bsmm_xcol_v2.h, GATED): forced on small layouts (gates 0, negative, > 1; dense phases; N not a
    multiple of the row tile) and at the bench shape, fprop and bprop against the oracle, which gates the fp32 block product."""
    torch, BSMM = env
    from blocksparse_amd import _lib
    cases = [(P.random_layout(9, 7, 0.4, seed=8), (72, 200)), (np.ones((5, 20), dtype=bool), (128,)), (P.ba_layout(40, 3, seed=1), (264,))]
    try:
        _lib.set_kernel_variant(3)
        for li, (lay, Ns) in enumerate(cases):
            b = BSMM(lay, block_size=32, feature_axis=axis)
            t = O.build_layout_luts(lay, 32)
            for N in Ns:
                W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), dtype, seed=60 + N)
                g = MG.gate_inputs(b.blocks, 70 + li)
                tg = torch.from_numpy(g).cuda()
                tw, tx, te = _t(torch, W, dtype), _t(torch, X, dtype), _t(torch, E, dtype)
                y = b.fprop(tx, tw, gate=tg)
                assert _lib.last_kernel() == _lib.K_XCOL32_STAGED
                dx = b.bprop(te, tw, gate=tg)
                assert _lib.last_kernel() == _lib.K_XCOL32_STAGED
                for name, got, ref in (("Y", y, O.fprop(t, X, W, axis, gate=g)), ("DX", dx, O.bprop(t, E, W, axis, gate=g))):
                    l2, _ = P.errors(got.float().cpu().numpy(), O.round_to(ref, dtype))
                    assert l2 <= P.L2_BAR[dtype], (axis, dtype, li, N, name, l2)
    finally:
        _lib.set_kernel_variant(0)
    # the bench shape, library's own choice of kernel: sampled output block columns against the oracle
    lay = P.random_layout(128, 128, 0.2, seed=1234)
    b = BSMM(lay, block_size=32, feature_axis=axis)
    b.gate_images = False
    t = O.build_layout_luts(lay, 32)
    N = 4096
    W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), dtype, seed=7)
    g = MG.gate_inputs(b.blocks, 9)
    tg = torch.from_numpy(g).cuda()
    y = b.fprop(_t(torch, X, dtype), _t(torch, W, dtype), gate=tg)
    assert _lib.last_kernel() == _lib.K_XCOL32_STAGED
    yh = y.float().cpu().numpy()
    cols = [0, 37, 127]
    for k, ref in O.fprop_cols(t, X * 1.0, np.asarray(W, dtype=np.float64) * g[:, None, None], axis, cols).items():
        got = yh[:, k * 32:(k + 1) * 32] if axis else yh[k * 32:(k + 1) * 32, :]
        l2, _ = P.errors(got, O.round_to(ref, dtype))
        assert l2 <= P.L2_BAR[dtype], (axis, dtype, k, l2)


@pytest.mark.gpu
@pytest.mark.parametrize("bs", [32, 16, 8])
def test_gated_kernels_against_reference_fixtures(env, bs):
    torch, BSMM = env
    gold = np.load(os.path.join(HERE, "golden", "gate.npz"))
    key, lay, t, W, X, E, g = _case(gold, bs)
    b = BSMM(lay, block_size=bs, feature_axis=0)
    tg = torch.from_numpy(g).cuda()
    tw, tx, te = _t(torch, W, "f32"), _t(torch, X, "f32"), _t(torch, E, "f32")
    for name, got in (("Y", b.fprop(tx, tw, gate=tg)), ("DX", b.bprop(te, tw, gate=tg)), ("DW", b.updat(tx, te, gate=tg))):
        l2, _ = P.errors(got.cpu().numpy(), gold[key + name])
        assert l2 < 2e-6, (bs, name, l2)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_gate_grad_and_autograd(env, dtype):
    torch, BSMM = env
    lay = P.ba_layout(16, 2, seed=3)
    bs, N = 32, 96
    b = BSMM(lay, block_size=bs, feature_axis=1)
    t = O.build_layout_luts(lay, bs)
    W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), dtype, seed=77)
    g = MG.gate_inputs(b.blocks, 78)
    tg = torch.from_numpy(g).cuda().requires_grad_(True)
    tw, tx = _t(torch, W, dtype).requires_grad_(True), _t(torch, X, dtype).requires_grad_(True)
    y = b(tx, tw, gate=tg, gate_grad=True, dw_gated=False)
    y.backward(_t(torch, E, dtype))
    rnd = lambda a: O.round_to(a, dtype)
    DW = rnd(O.updat(t, X, E, 1))                               # ungated dw, rounded to storage as the kernel leaves it
    dw_ref, dg_ref = O.gate_grad(DW, W, g)
    for name, got, ref in (("Y", y, rnd(O.fprop(t, X, W, 1, gate=g))), ("DX", tx.grad, rnd(O.bprop(t, E, W, 1, gate=g))),
                           ("DW", tw.grad, rnd(dw_ref)), ("DG", tg.grad, dg_ref)):
        l2, _ = P.errors(got.detach().float().cpu().numpy(), ref)
        assert l2 <= max(P.L2_BAR[dtype], 2e-6), (dtype, name, l2)
    # dw_gated without gate_grad: dw scaled inside updat, no dg
    tw2 = _t(torch, W, dtype).requires_grad_(True)
    b(tx.detach(), tw2, gate=tg.detach(), dw_gated=True).backward(_t(torch, E, dtype))
    l2, _ = P.errors(tw2.grad.float().cpu().numpy(), rnd(O.updat(t, X, E, 1, gate=g)))
    assert l2 <= P.L2_BAR[dtype], l2
    with pytest.raises(ValueError):
        b.fprop(tx.detach(), tw.detach(), gate=tg.detach()[:-1])


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("bs,axis", [(32, 1), (32, 0), (16, 0), (16, 1)])
def test_gated_calls_run_the_ungated_kernels(env, bs, axis, dtype):
    """Round 6: a gated 16-bit fprop / bprop with a long minibatch = bsmm_gate_weights + the UNGATED call (the reference gates inside its main
    tensor-core kernels, src/blocksparse_hgemm_cn_64_op_gpu.cu:54-66,96-124; here the fast kernels carry no gate logic and run over gated weight
    images instead): 0 / 1 gates -> one exact image over the plain tables, other gates -> hi / lo images over the doubled tables.  Every output
    element against the oracle (which gates the float64 block product); the kernel family that ran is the ungated call's; a gate-0 block
    holding Inf contributes nothing; the in-kernel GATED path (gate_images = False) agrees."""
    torch, BSMM = env
    from blocksparse_amd import _lib
    lay = P.random_layout(40, 24, 0.3, seed=21) if bs == 32 else P.random_layout(72, 56, 0.15, seed=22)
    b = BSMM(lay, block_size=bs, feature_axis=axis)
    b.GATE_IMAGES_MIN_N = {32: 1024, 16: 1024}      # (bsize 16 takes the images from 2048 rows by default: a measured rule, not a limit)
    t = O.build_layout_luts(lay, bs)
    rs = np.random.RandomState(5)
    for N in (1024, 1160):
        W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), dtype, seed=30 + N)
        tw, tx, te = _t(torch, W, dtype), _t(torch, X, dtype), _t(torch, E, dtype)
        y0 = b.fprop(tx, tw)
        fam_f = _lib.last_kernel()
        b.bprop(te, tw)
        fam_b = _lib.last_kernel()
        for kind in ("binary", "general"):
            g = (rs.rand(b.blocks) < 0.8).astype(np.float32) if kind == "binary" else MG.gate_inputs(b.blocks, 40 + N)
            tg = torch.from_numpy(g).cuda()
            assert b._gate_kind_of(tg) == kind
            general2 = kind == "general" and dtype == "bf16"      # (fp16: one image for any gate, the reference's own rounding of g w)
            y = b.fprop(tx, tw, gate=tg)
            assert general2 or _lib.last_kernel() == fam_f, (kind, _lib.last_kernel(), fam_f)    # (twice the blocks: the cost model
            dx = b.bprop(te, tw, gate=tg)                                                          #  may choose another ungated kernel)
            assert general2 or _lib.last_kernel() == fam_b, (kind, _lib.last_kernel(), fam_b)
            for name, got, ref in (("Y", y, O.fprop(t, X, W, axis, gate=g)), ("DX", dx, O.bprop(t, E, W, axis, gate=g))):
                l2, _ = P.errors(got.float().cpu().numpy(), O.round_to(ref, dtype))
                assert l2 <= P.L2_BAR[dtype], (bs, axis, dtype, N, kind, name, l2)
            # a block that is gated off may hold anything
            off = np.nonzero(g == 0)[0]
            W2 = np.array(W, dtype=np.float32, copy=True)
            W2[off[0]] = np.inf
            y2 = b.fprop(tx, _t(torch, W2, dtype), gate=tg)
            assert torch.equal(y2, y)
            # the GATED staged kernels compute the same thing (their own summation order)
            b.gate_images = False
            try:
                ys = b.fprop(tx, tw, gate=tg)
            finally:
                b.gate_images = True
            l2, _ = P.errors(ys.float().cpu().numpy(), y.float().cpu().numpy())
            assert l2 <= 2 * P.L2_BAR[dtype], (kind, l2)
        assert torch.equal(b.fprop(tx, tw, gate=torch.ones(b.blocks, device="cuda")), y0)     # all-ones gate: the ungated call on a copy of W


@pytest.mark.gpu
def test_gated_flow_kernel_at_the_bench_shape(env):
    """a gated bsize-32 / feature-axis-1 call at the bench shape runs the flow kernel (BSMM_K_XCOL32_FLOW), for a pruning
    mask and for arbitrary gates; sampled output block columns against the oracle."""
    torch, BSMM = env
    from blocksparse_amd import _lib
    lay = P.random_layout(128, 128, 0.2, seed=1234)
    b = BSMM(lay, block_size=32, feature_axis=1)
    t = O.build_layout_luts(lay, 32)
    N = 4096
    W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), "bf16", seed=7)
    rs = np.random.RandomState(3)
    for g in ((rs.rand(b.blocks) < 0.8).astype(np.float32), MG.gate_inputs(b.blocks, 9)):
        y = b.fprop(_t(torch, X, "bf16"), _t(torch, W, "bf16"), gate=torch.from_numpy(g).cuda())
        assert _lib.last_kernel() == _lib.K_XCOL32_FLOW
        yh = y.float().cpu().numpy()
        for k, ref in O.fprop_cols(t, X * 1.0, np.asarray(W, dtype=np.float64) * g[:, None, None], 1, [0, 37, 127]).items():
            l2, _ = P.errors(yh[:, k * 32:(k + 1) * 32], O.round_to(ref, "bf16"))
            assert l2 <= P.L2_BAR["bf16"], (k, l2)
