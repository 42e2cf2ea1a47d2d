"""GPU parity of the block-sparse attention path (through the C ABI of include/bst.h) against oracle/bst_oracle.py
and the fixtures generated from the reference.  Bars: results are compared with the float64 oracle evaluated on the
SAME (exactly representable) inputs and rounded ONCE to the storage type: L2-relative <= 1e-3 for 16-bit outputs
(north_star), <= 2e-6 for fp32 outputs; masks are bit-exact."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden_bst as G
from oracle import bst_oracle as O
from oracle import bsmm_oracle as R          # rounding helpers

pytestmark = pytest.mark.gpu
L2 = {"f32": 2e-6, "f16": 1e-3, "bf16": 1e-3}


@pytest.fixture(scope="module")
def env():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import __graft_entry__ as g
    g.build()
    from blocksparse_amd import BlocksparseTransformer
    return torch, BlocksparseTransformer


def _tt(torch, a, dt):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to("cuda").to(getattr(torch, {"f32": "float32", "f16": "float16", "bf16": "bfloat16"}[dt]))


def _np(t):
    return t.detach().float().cpu().numpy().astype(np.float64)


def _err(got, ref):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30)


def _run_case(torch, BST, lay, heads, bsize, hs, batch, cb, seed, act, score):
    """All five ops on one configuration; returns {name: l2 error}."""
    bst = BST(lay, block_size=bsize, heads=heads, mask_callback=cb)
    L = O.build_luts(lay)
    inp = G.gen_inputs(np.asarray(lay), heads, bsize, hs, batch, bst.blocks, seed)
    rq = lambda a: R.round_to(a, act)                                   # activations as the device will hold them
    rs = lambda a: R.round_to(a, score)
    Q, K, V, E = rq(inp["Q"]), rq(inp["K"]), rq(inp["V"]), rq(inp["E"])
    W, X, DY = rs(inp["W"]), rs(inp["X"]), rs(inp["DY"])
    scale = 1.0 / np.sqrt(hs)
    out = {}
    tq, tk, tv, te = (_tt(torch, a, act) for a in (Q, K, V, E))
    tw, tx, tdy = (_tt(torch, a, score) for a in (W, X, DY))
    sdt = tw.dtype
    out["NT"] = _err(_np(bst._nt(tq, tk, sdt)), rs(O.nt(L, Q, K, bsize, heads)))
    out["NN"] = _err(_np(bst._xn(tw, tv, False)), rq(O.nn(L, W, V, bsize, heads)))
    out["TN"] = _err(_np(bst._xn(tw, te, True)), rq(O.tn(L, W, E, bsize, heads)))
    mask_np = bst.softmax_mask_np
    mask_t = bst._table("mask", "cuda") if cb else None
    Yref = O.masked_softmax(L, X, bsize, scale, mask_np)
    ty = bst._softmax_fwd(tx, scale, mask_t, sdt)
    out["SM"] = _err(_np(ty), rs(Yref))
    Yr = rs(Yref)
    out["SMG"] = _err(_np(bst._softmax_bwd(tdy, _tt(torch, Yr, score), scale)), rs(O.masked_softmax_grad(L, DY, Yr, scale)))
    return out


@pytest.mark.parametrize("case", G.MATH_CASES, ids=[c[0] for c in G.MATH_CASES])
@pytest.mark.parametrize("act,score", [("f32", "bf16"), ("f16", "f16"), ("bf16", "bf16")])
def test_all_ops_against_oracle(env, case, act, score):
    torch, BST = env
    name, lkey, heads, bsize, hs, batch, cbn, seed = case
    res = _run_case(torch, BST, G.layouts()[lkey], heads, bsize, hs, batch, G.CALLBACKS[cbn], seed, act, score)
    for k, v in res.items():
        bar = L2[score] if k in ("NT", "SM", "SMG") else L2[act]
        assert v < bar, (name, act, score, k, v)


@pytest.mark.parametrize("case", G.MATH_CASES, ids=[c[0] for c in G.MATH_CASES])
def test_against_reference_fixtures(env, case):
    """Device results vs the numbers the reference's own NumPy oracle produced (fp32 activations, bf16 scores: the
    reference's fp32 pathway).  Inputs are fp16-representable, so the only differences are the bf16 storage of
    scores / softmax outputs (2^-9 relative)."""
    torch, BST = env
    gold = np.load(os.path.join(HERE, "golden", "bst.npz"))
    name, lkey, heads, bsize, hs, batch, cbn, seed = case
    lay = G.layouts()[lkey]
    bst = BST(lay, block_size=bsize, heads=heads, mask_callback=G.CALLBACKS[cbn])
    inp = G.gen_inputs(lay, heads, bsize, hs, batch, bst.blocks, seed)
    key = "math/%s/" % name
    scale = float(gold[key + "scale"])
    q, k, v, e = (_tt(torch, inp[n], "f32") for n in ("Q", "K", "V", "E"))
    w, x, dy = (_tt(torch, inp[n], "bf16") for n in ("W", "X", "DY"))      # fp16-representable values of |x| < 8 are not all bf16-exact
    assert _err(G.sub(_np(bst.nt_op(q, k))), gold[key + "NT"]) < 4e-3
    bar_w = 4e-3                                                           # W rounded to bf16 on the way in
    assert _err(G.sub(_np(bst.nn_op(w, v))), gold[key + "NN"]) < bar_w
    assert _err(G.sub(_np(bst.tn_op(w, e))), gold[key + "TN"]) < bar_w
    y = bst.masked_softmax(x, scale=scale)
    assert _err(G.sub(_np(y)), gold[key + "SM"]) < 6e-3
    if key + "SM_AR" in gold.files:
        yar = bst.masked_softmax(x, scale=scale, autoregress_at_key=int(gold[key + "akey"]))
        assert _err(G.sub(_np(yar)), gold[key + "SM_AR"]) < 6e-3


@pytest.mark.parametrize("bsize", [8, 16, 32, 64])
def test_partial_autoregressive_mask_bit_exact(env, bsize):
    torch, BST = env
    lay = G.layouts()["rect_3heads"]
    bst = BST(lay, block_size=bsize, mask_callback=G.head_cb)
    L = O.build_luts(lay)
    for key in (0, bsize // 2, 3 * bsize + 1, 10 * bsize - 1):
        got = bst.partial_autoregressive_mask(key, "cuda").cpu().numpy().view(O.mask_dtype(bsize))
        ref = O.partial_autoregressive_mask(bst.softmax_mask, L["nt_lut"], bsize, key)
        assert np.array_equal(got, ref), (bsize, key)
    with pytest.raises(ValueError):
        bst.partial_autoregressive_mask(10 * bsize, "cuda")


def test_ragged_layouts_and_head_state(env):
    """Empty query rows / key columns (outputs must be zero there), a single block, head_state not a multiple of 32."""
    torch, BST = env
    lay = np.zeros((2, 5, 7), dtype=np.int32)
    lay[0, 1, 2] = lay[0, 1, 6] = lay[0, 4, 0] = 1
    lay[1, 0, 0] = lay[1, 3, 3] = lay[1, 3, 4] = 1
    for bsize, hs in ((32, 24), (64, 40), (16, 8), (8, 16), (32, 96), (32, 160), (64, 136), (32, 32), (32, 128), (64, 64)):   # direct and LDS-DMA (1 / 2 / 4 chunk) nt kernels
        res = _run_case(torch, BST, lay, 2, bsize, hs, 2, G.head_cb, 5, "f32", "bf16")
        assert res["NT"] < 1e-3 and res["SM"] < 1e-3 and res["SMG"] < 1e-3 and res["NN"] < 2e-6 and res["TN"] < 2e-6, (bsize, hs, res)
    one = np.ones((1, 1, 1), dtype=np.int32)
    res = _run_case(torch, BST, one, 3, 32, 64, 1, None, 6, "f32", "bf16")
    assert max(res["NN"], res["TN"]) < 2e-6 and max(res["NT"], res["SM"], res["SMG"]) < 1e-3, res


@pytest.mark.parametrize("act", ["f16", "bf16"])
def test_native_16bit_mfma_kernels(env, act):
    """16-bit activations with scores of the same type run on v_mfma_f32_32x32x16 without widening (LDS-DMA tiles +
    transposing reads); head sizes 32 / 64 / 128 take those kernels, 96 the widening ones; bsize 64 = 2 x 2 tiles."""
    torch, BST = env
    lay = G.layouts()["rect_3heads"]
    for bsize, hs in ((32, 64), (32, 32), (32, 128), (32, 96), (64, 64), (64, 32)):
        res = _run_case(torch, BST, lay, 3, bsize, hs, 2, G.head_cb, 17 + hs, act, act)
        assert max(res.values()) < 1e-3, (act, bsize, hs, res)
    tri = G.layouts()["causal_2heads"]
    res = _run_case(torch, BST, tri, 2, 32, 64, 3, G.causal_cb, 23, act, act)
    assert max(res.values()) < 1e-3, (act, res)


def test_fully_masked_row_is_uniform(env):
    """A query row with no visible key: the reference's NumPy oracle yields a uniform row; so do we (include/bst.h)."""
    torch, BST = env
    def cb(shape, h, q, k, b):
        m = np.ones(shape, dtype=bool)
        m[3, :] = False
        return m
    bst = BST(np.ones((1, 2, 2), dtype=np.int32), block_size=32, heads=1, mask_callback=cb)
    x = torch.randn(1, 1, 4, 32, 32, device="cuda", generator=torch.Generator(device="cuda").manual_seed(51)).bfloat16()
    y = _np(bst.masked_softmax(x, scale=0.5))
    assert np.allclose(y[0, 0, 0:2, 3, :], 1.0 / 64, rtol=1e-2) and np.all(np.isfinite(y))
    ref = bst.masked_softmax_test(_np(x).astype(np.float32), scale=0.5)
    assert _err(y, ref) < 6e-3


def test_autograd_chain_matches_reference_gradients(env):
    """y = nn(softmax(nt(q, k)), v): dq, dk, dv through the registered gradients vs the oracle chain
    (test/blocksparse_transformer_test.py:150-182)."""
    torch, BST = env
    lay = G.layouts()["causal_2heads"]
    heads, bsize, hs, batch = 2, 32, 64, 2
    bst = BST(lay, block_size=bsize, mask_callback=G.causal_cb)
    L = O.build_luts(lay)
    inp = G.gen_inputs(lay, heads, bsize, hs, batch, bst.blocks, 21)
    scale = 1.0 / np.sqrt(hs)
    q, k, v = (_tt(torch, inp[n], "f32").requires_grad_(True) for n in ("Q", "K", "V"))
    e = _tt(torch, inp["E"], "f32")
    w = bst.query_key_op(q, k)
    a = bst.masked_softmax(w, scale=scale)
    y = bst.weight_value_op(a, v)
    y.backward(e)
    bf = lambda t: R.round_to(t, "bf16")
    W = bf(O.nt(L, inp["Q"], inp["K"], bsize, heads))
    A = bf(O.masked_softmax(L, W, bsize, scale, bst.softmax_mask_np))
    Y = O.nn(L, A, inp["V"], bsize, heads)
    DV = O.tn(L, A, inp["E"], bsize, heads)
    DA = bf(O.nt(L, inp["E"], inp["V"], bsize, heads))
    DW = bf(O.masked_softmax_grad(L, DA, A, scale))
    DQ, DK = O.nn(L, DW, inp["K"], bsize, heads), O.tn(L, DW, inp["Q"], bsize, heads)
    assert _err(_np(y), Y) < 3e-3 and _err(_np(v.grad), DV) < 3e-3
    assert _err(_np(q.grad), DQ) < 1e-2 and _err(_np(k.grad), DK) < 1e-2      # three bf16 roundings deep; an ulp flip upstream moves a row


def test_baseline_config_properties(env):
    """BASELINE configs[4] at full size (batch 4, 16 heads x 64, ctx 4096, bsize 32, local+strided causal layout):
    size-independent properties.  softmax rows sum to 1 and vanish above the diagonal; nt is linear in q; nn with
    all-ones scores equals the block-row sums of v."""
    torch, BST = env
    lay = O.local_strided_layout(128)
    bst = BST(lay, block_size=32, heads=16, mask_callback=O.causal_mask_callback)
    assert bst.blocks == 1466 and bst.nn_max == 19
    g = torch.Generator(device="cuda").manual_seed(1)
    shp = (4, 4096, 1024)
    q1 = torch.rand(shp, device="cuda", generator=g) * 2 - 1
    q2 = torch.rand(shp, device="cuda", generator=g) * 2 - 1
    k = torch.rand(shp, device="cuda", generator=g) * 2 - 1
    w1, w2, w12 = bst.nt_op(q1, k).float(), bst.nt_op(q2, k).float(), bst.nt_op(q1 + q2, k).float()
    assert ((w1 + w2 - w12).norm() / w12.norm()).item() < 6e-3                     # three bf16 roundings
    a = bst.masked_softmax(bst.nt_op(q1, k), scale=0.125).float()
    nn_lut = bst.nn_lut[0]
    rows = torch.zeros(4, 16, 128, 32, device="cuda")
    ids = torch.from_numpy(np.repeat(np.arange(128), nn_lut[:128, 1])).to("cuda")
    rows.index_add_(2, ids, a.sum(dim=-1)[:, :, torch.from_numpy(nn_lut[128:, 0].astype(np.int64)).to("cuda")])
    assert (rows - 1).abs().max().item() < 2e-2                                    # sums of <= 608 bf16 values
    diag = torch.from_numpy(np.nonzero(bst.nt_lut[0][:, 0] == bst.nt_lut[0][:, 1])[0]).to("cuda")
    assert a[:, :, diag].triu(1).abs().max().item() == 0.0
    ones = torch.ones(4, 16, 1466, 32, 32, device="cuda", dtype=torch.bfloat16)
    y = bst.nn_op(ones, k)
    kb = k.view(4, 128, 32, 16, 64).sum(dim=2)                                      # per key block
    exp = torch.zeros(4, 128, 16, 64, device="cuda")
    exp.index_add_(1, ids, kb[:, torch.from_numpy(nn_lut[128:, 1].astype(np.int64)).to("cuda")])
    got = y.view(4, 128, 32, 16, 64)
    assert ((got - exp[:, :, None]).norm() / exp.norm() / np.sqrt(32)).item() < 1e-5


# ---- scores + softmax in one launch (bst_nt_softmax, round 6) -----------------------------------------------------------
def _fused_case(torch, BST, lay, heads, hs, batch, cb, seed, act, score, akey=None):
    """query_key_softmax against the float64 oracle chain nt -> round to the score type -> masked softmax, and against the two-launch device path."""
    import _parity as P
    bst = BST(lay, block_size=32, heads=heads, mask_callback=cb)
    L = O.build_luts(lay)
    inp = G.gen_inputs(np.asarray(lay), heads, 32, hs, batch, bst.blocks, seed)
    Q, K = R.round_to(inp["Q"], act), R.round_to(inp["K"], act)
    scale = 1.0 / np.sqrt(hs)
    tq, tk = _tt(torch, Q, act), _tt(torch, K, act)
    sdt = getattr(torch, {"f16": "float16", "bf16": "bfloat16"}[score])
    mask_t = None
    if cb is not None:
        mask_t = bst.partial_autoregressive_mask(akey, "cuda") if akey is not None else bst._table("mask", "cuda")
    fused = bst._nt_softmax(tq, tk, scale, mask_t, sdt)
    two = bst._softmax_fwd(bst._nt(tq, tk, sdt), scale, mask_t, sdt)
    W = R.round_to(O.nt(L, Q, K, 32, heads), score)
    mask_np = bst.softmax_mask_np
    if akey is not None:
        mask_np = np.ascontiguousarray(O.partial_autoregressive_mask(bst.softmax_mask, L["nt_lut"], 32, akey).transpose(0, 2, 1))
    Y = O.masked_softmax(L, W, 32, scale, mask_np)
    return bst, fused, two, Y


@pytest.mark.parametrize("act,score", [("f32", "bf16"), ("bf16", "bf16"), ("f16", "f16")])
@pytest.mark.parametrize("hs", [32, 64, 128])
def test_fused_scores_softmax_against_oracle(env, act, score, hs):
    """Local + strided causal layout (the pattern of BASELINE configs[4], 32 context blocks: rows of 1 .. 7 blocks), a rectangular layout
    with empty rows, head-dependent masks, a partial autoregressive mask: every block against the oracle chain, and within one step of the
    score type of the two-launch path element by element (the same rounded scores enter the same softmax arithmetic)."""
    import _parity as P
    torch, BST = env
    cases = [(O.local_strided_layout(32), 2, 2, O.causal_mask_callback, None),
             (O.local_strided_layout(32), 2, 1, O.causal_mask_callback, 7 * 32 + 5),
             (G.layouts()["rect_3heads"], 3, 2, G.head_cb, None),
             (np.ones((1, 3, 20), dtype=np.int32), 2, 1, None, None)]              # rows of exactly 20 blocks: five tiles per wave
    for ci, (lay, heads, batch, cb, akey) in enumerate(cases):
        bst, fused, two, Y = _fused_case(torch, BST, lay, heads, hs, batch, cb, 30 + ci, act, score, akey)
        assert fused is not None, (ci, "the fused kernel serves this configuration")
        got = _np(fused)
        assert np.isfinite(got).all()
        assert _err(got, R.round_to(Y, score)) < L2[score], (ci, act, score, hs, _err(got, R.round_to(Y, score)))
        rep = P.block_report(got.reshape(-1, 32 * 32), _np(two).reshape(-1, 32 * 32), score, got.size // 1024)
        assert rep["elem_bad"] == 0 and rep["tensor_l2"] < 2e-4, (ci, act, score, hs, rep)      # against the two launches: summation order only
        # blocks of empty query rows do not exist; every stored block's rows sum to one over the row's blocks
    lay = np.ones((1, 2, 21), dtype=np.int32)                                      # 21 blocks in a row: not served, the operator composes the two
    bst = BST(lay, block_size=32, heads=1)
    q = _tt(torch, np.zeros((1, 64, hs)), act)
    k = _tt(torch, np.zeros((1, 21 * 32, hs)), act)
    assert bst._nt_softmax(q, k, 1.0, None, getattr(torch, {"f16": "float16", "bf16": "bfloat16"}[score])) is None
    y = bst.query_key_softmax(q, k, scale=1.0)
    assert np.allclose(_np(y), 1.0 / (21 * 32), rtol=1e-2)


def test_fused_operator_gradients_match_the_composed_operators(env):
    """query_key_softmax(q, k) under autograd against masked_softmax(query_key_op(q, k)): the same forward values within a step of bf16 and
    gradients that agree to the bar of test_autograd_chain_matches_reference_gradients."""
    torch, BST = env
    lay = O.local_strided_layout(16, local=3, stride=4)
    heads, hs, batch = 2, 64, 2
    bst = BST(lay, block_size=32, heads=heads, mask_callback=O.causal_mask_callback)
    inp = G.gen_inputs(lay, heads, 32, hs, batch, bst.blocks, 41)
    scale = 1.0 / np.sqrt(hs)
    e = _tt(torch, inp["E"], "f32")
    outs = []
    for fused in (True, False):
        q, k, v = (_tt(torch, inp[n], "f32").requires_grad_(True) for n in ("Q", "K", "V"))
        a = bst.query_key_softmax(q, k, scale=scale) if fused else bst.masked_softmax(bst.query_key_op(q, k), scale=scale)
        y = bst.weight_value_op(a, v)
        y.backward(e)
        outs.append((_np(a), _np(y), _np(q.grad), _np(k.grad), _np(v.grad)))
    for name, f, c, bar in zip(("a", "y", "dq", "dk", "dv"), outs[0], outs[1], (2e-4, 1e-3, 1e-2, 1e-2, 1e-3)):
        assert _err(f, c) < bar, (name, _err(f, c))


def test_fused_at_baseline_config(env):
    """BASELINE configs[4] at full size: the fused operator's rows sum to one, vanish above the diagonal, and equal the two launches within a bf16 step."""
    import _parity as P
    torch, BST = env
    lay = O.local_strided_layout(128)
    bst = BST(lay, block_size=32, heads=16, mask_callback=O.causal_mask_callback)
    g = torch.Generator(device="cuda").manual_seed(2)
    shp = (4, 4096, 1024)
    q = torch.rand(shp, device="cuda", generator=g) * 2 - 1
    k = torch.rand(shp, device="cuda", generator=g) * 2 - 1
    a = bst.query_key_softmax(q, k, scale=0.125)
    b2 = bst.masked_softmax(bst.query_key_op(q, k), scale=0.125)
    diff = (a.float() - b2.float()).abs()
    assert (diff.norm() / b2.float().norm()).item() < 2e-4
    step = torch.maximum(a.float().abs(), b2.float().abs()) * 2.0 ** -7 + 1e-6     # one bf16 step at the value's binade, generously
    assert bool((diff <= step).all())
    af = a.float()
    nn_lut = bst.nn_lut[0]
    rows = torch.zeros(4, 16, 128, 32, device="cuda")
    ids = torch.from_numpy(np.repeat(np.arange(128), nn_lut[:128, 1])).to("cuda")
    rows.index_add_(2, ids, af.sum(dim=-1)[:, :, torch.from_numpy(nn_lut[128:, 0].astype(np.int64)).to("cuda")])
    assert (rows - 1).abs().max().item() < 2e-2
    diag = torch.from_numpy(np.nonzero(bst.nt_lut[0][:, 0] == bst.nt_lut[0][:, 1])[0]).to("cuda")
    assert af[:, :, diag].triu(1).abs().max().item() == 0.0


@pytest.mark.parametrize("act,score", [("f32", "bf16"), ("bf16", "bf16"), ("f16", "f16")])
def test_fused_backward_pair_against_oracle(env, act, score):
    """bst_nt_softmax_grad: dx = softmax_grad(round(e . v^T), probs) as one launch, against the float64 oracle chain and within one step of the
    score type of the two launches (bst_nt, bst_softmax_grad)."""
    import _parity as P
    torch, BST = env
    for ci, (lay, heads, batch, hs) in enumerate(((O.local_strided_layout(32), 2, 2, 64), (G.layouts()["rect_3heads"], 3, 1, 32),
                                                  (np.ones((1, 2, 20), dtype=np.int32), 2, 1, 128))):
        bst = BST(lay, block_size=32, heads=heads, mask_callback=O.causal_mask_callback if ci == 0 else None)
        L = O.build_luts(lay)
        inp = G.gen_inputs(np.asarray(lay), heads, 32, hs, batch, bst.blocks, 50 + ci)
        E, V = R.round_to(inp["E"], act), R.round_to(inp["V"], act)
        X = R.round_to(inp["X"], score)
        scale = 1.0 / np.sqrt(hs)
        Yp = R.round_to(O.masked_softmax(L, X, 32, scale, bst.softmax_mask_np), score)          # some probabilities
        te, tv, ty = _tt(torch, E, act), _tt(torch, V, act), _tt(torch, Yp, score)
        fused = bst._nt_softmax_grad(te, tv, ty, scale)
        assert fused is not None
        two = bst._softmax_bwd(bst._nt(te, tv, ty.dtype), ty, scale)
        DP = R.round_to(O.nt(L, E, V, 32, heads), score)
        ref = O.masked_softmax_grad(L, DP, Yp, scale)
        got = _np(fused)
        assert np.isfinite(got).all()
        assert _err(got, R.round_to(ref, score)) < L2[score], (ci, act, score, _err(got, R.round_to(ref, score)))
        rep = P.block_report(got.reshape(-1, 1024), _np(two).reshape(-1, 1024), score, got.size // 1024)
        assert rep["tensor_l2"] < 3e-4, (ci, act, score, rep)


def test_attention_operator_matches_the_composed_operators(env):
    """BlocksparseTransformer.attention(q, k, v): values and all three gradients against the three composed operators."""
    torch, BST = env
    lay = O.local_strided_layout(16, local=3, stride=4)
    heads, hs, batch = 2, 64, 2
    bst = BST(lay, block_size=32, heads=heads, mask_callback=O.causal_mask_callback)
    inp = G.gen_inputs(lay, heads, 32, hs, batch, bst.blocks, 43)
    scale = 1.0 / np.sqrt(hs)
    e = _tt(torch, inp["E"], "f32")
    outs = []
    for fused in (True, False):
        q, k, v = (_tt(torch, inp[n], "f32").requires_grad_(True) for n in ("Q", "K", "V"))
        y = bst.attention(q, k, v, scale=scale) if fused else bst.weight_value_op(bst.masked_softmax(bst.query_key_op(q, k), scale=scale), v)
        y.backward(e)
        outs.append((_np(y), _np(q.grad), _np(k.grad), _np(v.grad)))
    for name, f, c, bar in zip(("y", "dq", "dk", "dv"), outs[0], outs[1], (1e-3, 1e-2, 1e-2, 1e-3)):
        assert _err(f, c) < bar, (name, _err(f, c))
    # a layout the fused kernels do not serve (bsize 64): the operator composes the reference's three
    bst64 = BST(np.tril(np.ones((4, 4), dtype=np.int32)), block_size=64, heads=2)
    q, k, v = (torch.rand(1, 256, 128, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3 + i)).requires_grad_(True) for i in range(3))
    y = bst64.attention(q, k, v, scale=0.125)
    y.sum().backward()
    assert torch.isfinite(y).all() and torch.isfinite(q.grad).all() and torch.isfinite(k.grad).all() and torch.isfinite(v.grad).all()
