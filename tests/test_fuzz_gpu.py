"""Randomised parity tier: random block grids (1..70 per side; one seed up to 200), densities 3 % .. 100 %, both feature axes, bsize
32 / 16 / 8, bf16 / f16, ragged minibatches, per-block gates on a third of the bsize-32 cases, with the PLAN kernels forced
(BSMM_FLAG_FORCE_PLAN: the grouped / staged xprop kernels, the streaming / windowed updat kernels, the bsize-8 super-block path)
whatever the size heuristics would choose -- every output element against the float64 oracle (oracle/bsmm_oracle.py,
restating blocksparse/matmul.py:353-419), L2 bar 1e-3.  This is the only randomised coverage of the plan builders
(csrc/bsmm_plan.h); it used to be a lease-side script."""
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


def _one_case(torch, BSMM, lib, rng, it, big):
    bs = int(rng.choice([32, 32, 16, 8]))
    hi = 200 if big else 70
    CB, KB = int(rng.integers(1, hi)), int(rng.integers(1, hi))
    if bs == 8:                                   # the super-block path wants whole 32-feature blocks; others run the V_FMA kernels
        CB, KB = 4 * max(1, CB // 4), 4 * max(1, KB // 4)
    dens = float(rng.choice([0.03, 0.1, 0.2, 0.5, 1.0]))
    lay = rng.random((CB, KB)) < dens
    lay[rng.integers(0, CB), rng.integers(0, KB)] = True
    axis = int(rng.integers(0, 2))
    dtype = str(rng.choice(["bf16", "f16"]))
    N = int(rng.choice([8, 40, 128, 200, 264, 520] + ([1032, 2056, 4096] if big else [])))
    if axis == 0:
        N = (N + 7) // 8 * 8
    b = BSMM(lay, block_size=bs, feature_axis=axis)
    t = orc.build_layout_luts(lay, bs)
    W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), dtype, seed=it)
    w, x, e = P.to_dev(W, dtype, torch), P.to_dev(X, dtype, torch), P.to_dev(E, dtype, torch)
    use_gate = bs == 32 and rng.random() < 0.3
    g = (rng.random(b.blocks).astype(np.float32) * 2 - 0.5) if use_gate else None
    if g is not None:
        g[rng.random(b.blocks) < 0.3] = 0
        g[rng.random(b.blocks) < 0.3] = 1
    tg = torch.from_numpy(g).cuda() if g is not None else None
    y = P.to_host(b.fprop(x, w, gate=tg))
    kf = lib.last_kernel()
    dx = P.to_host(b.bprop(e, w, gate=tg))
    dw = P.to_host(b.updat(x, e))
    ku = lib.last_kernel()
    errs = (P.errors(y, orc.round_to(orc.fprop(t, X, W, axis, gate=g), dtype))[0],
            P.errors(dx, orc.round_to(orc.bprop(t, E, W, axis, gate=g), dtype))[0],
            P.errors(dw, orc.round_to(orc.updat(t, X, E, axis), dtype))[0])
    ctx = "bs%d %dx%d d%.2f axis%d %s N%d gate%d kernels %d/%d" % (bs, CB, KB, dens, axis, dtype, N, use_gate, kf, ku)
    assert np.isfinite(y).all() and np.isfinite(dx).all() and np.isfinite(dw).all(), ctx
    assert max(errs) <= P.L2_BAR[dtype], (ctx, errs)
    return kf, ku


@pytest.mark.parametrize("seed,ncase,big", [(0, 25, False), (1, 25, False), (2, 25, False), (7, 10, True)])
def test_fuzz_forced_plan_kernels(env, seed, ncase, big):
    torch, BSMM, lib = env
    rng = np.random.default_rng(seed)
    lib.set_kernel_variant(3)                      # FLAG_FORCE_PLAN on every call of the host classes
    seen = set()
    try:
        for it in range(ncase):
            seen.update(_one_case(torch, BSMM, lib, rng, it, big))
    finally:
        lib.set_kernel_variant(0)
    # the cases really ran plan kernels (not a generic fall-back for every one of them)
    assert seen & {lib.K_XCOL32_STAGED, lib.K_XCOL32_FLOW, lib.K_XCOL16_STAGED, lib.K_XPROP_SUPER8}, seen
    assert seen & {lib.K_UPDAT_STREAM, lib.K_UPDAT16_WIN, lib.K_UPDAT16_ROWS, lib.K_UPDAT_SUPER8}, seen
