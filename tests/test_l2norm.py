"""Block-sparse L2 weight norm: oracle + host NumPy methods against fixtures generated from the
reference (tests/golden/make_golden.py l2); GPU: kernels vs oracle incl. gain, mixed dtypes, the epsilon branch, autograd."""
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

CASES = [("holes", 32), ("holes", 8), ("rect", 16)]


def _case(gold, name, bs):
    key = "%s/bs%d/" % (name, bs)
    lay = gold[key + "layout"].astype(np.int32)
    seed = int(gold[key + "meta"][1])
    t = O.build_layout_luts(lay, bs)
    W, _, _ = MG.cfg0_inputs((t["blocks"], bs, bs), (1, 1), (1, 1), seed)
    W[0, :, 0] = 0.0
    U = np.random.RandomState(seed + 1).normal(0.0, 1.0, W.shape).astype(np.float16).astype(np.float32)
    return key, lay, t, W, U


@pytest.mark.parametrize("name,bs", CASES)
def test_oracle_and_host_methods_match_reference_fixtures(name, bs):
    from blocksparse_amd import BlocksparseMatMul
    gold = np.load(os.path.join(HERE, "golden", "l2norm.npz"))
    key, lay, t, W, U = _case(gold, name, bs)
    b = BlocksparseMatMul(lay, block_size=bs, feature_axis=0)
    Y, S = O.l2_normalize(t, W)
    D, _ = O.l2_normalize_grad(t, W, U)
    for what, got, ref in (("oracle Y", Y, gold[key + "Y"]), ("oracle DW", D, gold[key + "DW"]),
                           ("host Y", b.l2_normalize_test(W), gold[key + "Y"]), ("host DW", b.l2_normalize_grad_test(W, U), gold[key + "DW"])):
        err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        assert err < 1e-6, (key, what, err)
    # every output feature of the normalised weights has unit norm (or is all zero)
    Yd = O.to_dense(t, Y)
    n = np.sqrt((Yd ** 2).sum(axis=0))
    assert np.all((np.abs(n - 1) < 1e-9) | (n == 0))


@pytest.fixture(scope="module")
def env():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import __graft_entry__ as g
    g.build()
    from blocksparse_amd import BlocksparseMatMul
    return torch, BlocksparseMatMul


@pytest.mark.gpu
@pytest.mark.parametrize("xd,yd", [("f32", "f32"), ("bf16", "bf16"), ("f16", "f16"), ("f32", "bf16")])
@pytest.mark.parametrize("name,bs", CASES)
def test_kernels_match_oracle_and_fixtures(env, name, bs, xd, yd):
    torch, BSMM = env
    gold = np.load(os.path.join(HERE, "golden", "l2norm.npz"))
    key, lay, t, W, U = _case(gold, name, bs)
    W = O.round_to(W, xd)
    U = O.round_to(U, yd)
    b = BSMM(lay, block_size=bs, feature_axis=0)
    tt = lambda a, d: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda().to(getattr(torch, P.TORCH_DT[d]))
    rng = np.random.RandomState(5)
    gain = rng.uniform(0.5, 1.5, b.K).astype(np.float32)
    for g in (None, gain):
        tg = torch.from_numpy(g).cuda() if g is not None else None
        y, ss = b._l2_fwd(tt(W, xd), tg, 1e-12, getattr(torch, P.TORCH_DT[yd]))
        Y, S = O.l2_normalize(t, W, gain=g)
        l2, _ = P.errors(y.float().cpu().numpy(), O.round_to(Y, yd))
        assert l2 <= max(P.L2_BAR[yd], 2e-6), (key, xd, yd, "Y", l2)
        assert np.allclose(ss.cpu().numpy(), S, rtol=2e-6, atol=1e-30)
        dx, dg = b._l2_bwd(tt(U, yd), tt(W, xd), tg, ss, 1e-12)
        D, DG = O.l2_normalize_grad(t, W, U, gain=g)
        l2, _ = P.errors(dx.float().cpu().numpy(), O.round_to(D, xd))
        assert l2 <= max(P.L2_BAR[xd], 2e-6), (key, xd, yd, "DW", l2)
        if g is not None:
            l2, _ = P.errors(dg.cpu().numpy(), DG)
            assert l2 < 2e-6, (key, "DG", l2)
    if xd == "f32" and yd == "f32":                                    # the reference's numbers directly
        y, ss = b._l2_fwd(tt(W, xd), None, 1e-12, torch.float32)
        assert P.errors(y.cpu().numpy(), gold[key + "Y"])[0] < 2e-6
        dx, _ = b._l2_bwd(tt(U, yd), tt(W, xd), None, ss, 1e-12)
        assert P.errors(dx.cpu().numpy(), gold[key + "DW"])[0] < 2e-6


@pytest.mark.gpu
def test_epsilon_branch_and_autograd(env):
    torch, BSMM = env
    lay = np.eye(4, dtype=np.int32)                                   # every column block holds exactly one block
    b = BSMM(lay, block_size=8, feature_axis=0)
    t = O.build_layout_luts(lay, 8)
    rng = np.random.RandomState(9)
    W = rng.normal(size=b.w_shape).astype(np.float32)
    W[1] = 0.0                                                         # a whole column block below epsilon
    W[2, :, 3] = 1e-8                                                  # one feature below epsilon = 1e-12
    U = rng.normal(size=b.w_shape).astype(np.float32)
    gain = rng.uniform(0.5, 1.5, b.K).astype(np.float32)
    tw = torch.from_numpy(W).cuda().requires_grad_(True)
    tg = torch.from_numpy(gain).cuda().requires_grad_(True)
    y = b.l2_normalize(tw, gain=tg, epsilon=1e-12)
    y.backward(torch.from_numpy(U).cuda())
    Y, _ = O.l2_normalize(t, W, gain=gain)
    D, DG = O.l2_normalize_grad(t, W, U, gain=gain)
    assert np.all(np.isfinite(y.detach().cpu().numpy()))
    assert P.errors(y.detach().cpu().numpy(), Y)[0] < 2e-6
    assert P.errors(tw.grad.cpu().numpy(), D)[0] < 2e-6 and P.errors(tg.grad.cpu().numpy(), DG)[0] < 2e-6
    # the usual use: normalised weights feed the block-sparse matmul
    x = torch.randn(b.i_shape(16), device="cuda", generator=P.gen(torch, 41))
    out = b(x, b.l2_normalize(tw.detach()))
    assert torch.isfinite(out).all()
