"""GPU parity of the DIRECT blocks of the streaming weight-gradient kernel (bsize 32, feature axis 1, 16-bit types; 'BSU2' plans version 3,
csrc/bsmm_updat_v2.h::u2_direct_block, round 6): the blocks a 16 x 16-block window cannot hold in its 16 waves are multiplied by workgroups of
their own behind the schedule's -- one per quarter of the minibatch, the block's own 64-byte row pieces -- and meet in the summing pass.  Every
block of DW against the float64 oracle (oracle/bsmm_oracle.py::updat restating blocksparse/matmul.py:401-419 with the kernel semantics of alpha /
beta / pairs / gate, src/blocksparse_matmul_op_gpu.cu:2684-2814), the same calls on a plan WITHOUT direct blocks (PLAN_UPDAT_NO_DIRECT: overflow
items in a sliced last round, the form of rounds 2-5) beside it."""
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


def _crowded(CB, KB, dens, seed, extra):
    """random layout + `extra` more blocks in window (0, 0): a window with more blocks than slots"""
    lay = P.random_layout(CB, KB, dens, seed)
    rng = np.random.default_rng(seed + 1)
    free = np.argwhere(lay[:16, :16] == 0)
    for i in rng.permutation(len(free))[:extra]:
        lay[free[i][0], free[i][1]] = 1
    return lay


CASES = [
    # name, layout, N, dtype
    ("bench layout 20 % (two windows of 65 blocks), N 8192", P.random_layout(128, 128, 0.2, 1234), 8192, "bf16"),
    ("bench layout, ragged N 1000", P.random_layout(128, 128, 0.2, 1234), 1000, "f16"),
    ("bench layout, N 40: fewer chunks than parts x waves", P.random_layout(128, 128, 0.2, 1234), 40, "bf16"),
    ("bench layout, N 8: one chunk, three empty quarters", P.random_layout(128, 128, 0.2, 1234), 8, "bf16"),
    ("8192^2 5 % (BASELINE configs[3]): 32 x 32-block windows with direct blocks, N 1024", P.random_layout(256, 256, 0.05, 1234), 1024, "bf16"),
    ("48 x 40, window (0, 0) crowded to ~80 blocks, N 520", _crowded(48, 40, 0.2, 5, 30), 520, "bf16"),
    ("33 x 17 (ragged windows), crowded, N 2048", _crowded(33, 17, 0.22, 9, 25), 2048, "f16"),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_direct_blocks_against_the_oracle(env, case):
    torch, BSMM, lib = env
    name, lay, N, dt = case
    big = "32 x 32-block windows" in name                         # (the host class builds that plan for long minibatches only: force it here)
    b = BSMM(lay, block_size=32, feature_axis=1, plan_options=lib.PLAN_STREAM_32 if big else 0)
    bn = BSMM(lay, block_size=32, feature_axis=1, plan_options=lib.PLAN_UPDAT_NO_DIRECT | (lib.PLAN_STREAM_32 if big else 0))
    dev = torch.device("cuda")
    hp, hn = b._tables_on(dev).updat_plan.host, bn._tables_on(dev).updat_plan.host
    ndir = int(hp[28])
    assert int(hp[0]) == 0x42535532 and ndir > 0 and int(hn[28]) == 0 and int(hn[4]) > int(hp[4]), (name, ndir)
    direct_blocks = [int(hp[int(hp[29]) + 4 * d]) for d in range(ndir)]
    W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), dt, seed=13)
    x, e = P.to_dev(X, dt, torch), P.to_dev(E, dt, torch)
    dw0 = P.to_dev(np.random.default_rng(3).standard_normal(b.w_shape).astype(np.float32) * 0.05, dt, torch)
    gate = torch.rand(b.blocks, device="cuda", generator=P.gen(torch, 17)) * 2 - 0.5
    gate[direct_blocks[0]] = 0.0
    lib.set_kernel_variant(3)                  # the plan kernel whatever the minibatch
    try:
        got, ks = {}, []
        for tag, bb in (("direct", b), ("items", bn)):
            got[tag, "dw"] = P.to_host(bb.updat(x, e)); ks.append(lib.last_kernel())
            got[tag, "alpha / beta"] = P.to_host(bb.updat(x, e, alpha=0.5, beta=2.0, dw=dw0.clone())); ks.append(lib.last_kernel())
            got[tag, "two pairs"] = P.to_host(bb.updat([x, x], [e, e])); ks.append(lib.last_kernel())
            got[tag, "gated"] = P.to_host(bb.updat(x, e, gate=gate)); ks.append(lib.last_kernel())
            sums = bb.updat(x, e, sums_only=True); ks.append(lib.last_kernel())
            got[tag, "sums"] = sums.cpu().numpy().copy()
            got[tag, "sums + finalize"] = P.to_host(bb.updat_finalize(sums, alpha=0.5, beta=2.0, dw=dw0.clone(), gate=gate, dtype=x.dtype))
        again = P.to_host(b.updat(x, e))
    finally:
        lib.set_kernel_variant(0)
    assert set(ks) == {lib.K_UPDAT_STREAM}, ks
    t = orc.build_layout_luts(np.asarray(lay), 32)
    ref = orc.updat_fast(t, P.to_host(x).astype(np.float64), P.to_host(e).astype(np.float64), 1, dtype=np.float64)
    g64 = gate.cpu().numpy().astype(np.float64)[:, None, None]
    d64 = P.to_host(dw0).astype(np.float64)
    want = {"dw": ref, "alpha / beta": 0.5 * ref + 2.0 * d64, "two pairs": 2.0 * ref, "gated": ref * g64, "sums + finalize": 0.5 * g64 * ref + 2.0 * d64}
    for tag in ("direct", "items"):
        for what, w64 in want.items():
            P.assert_blocks(got[tag, what], w64, dt, b.blocks, (name, tag, what))
        P.assert_blocks(got[tag, "sums"], ref, "f32", b.blocks, (name, tag, "fp32 sums"))
    # the direct blocks themselves (a whole-tensor statement would not notice three blocks of 3 279)
    for w in direct_blocks:
        P.assert_blocks(got["direct", "dw"][w], ref[w], dt, 1, (name, "direct block", w))
    assert (got["direct", "gated"][direct_blocks[0]] == 0).all()
    assert np.array_equal(got["direct", "dw"], again)            # deterministic: the quarters are added in order


def test_direct_blocks_through_autograd_and_fp32(env):
    """The operator interface (w.grad) and the fp32 weight gradient (six bf16 piece pairs of ONE launch of the streaming kernel: the direct
    workgroups walk six pairs) on a plan with direct blocks."""
    torch, BSMM, lib = env
    lay = _crowded(48, 40, 0.2, 5, 30)
    b = BSMM(lay, block_size=32, feature_axis=1)
    assert int(b._tables_on(torch.device("cuda")).updat_plan.host[28]) > 0
    N = 1024
    t = orc.build_layout_luts(np.asarray(lay), 32)
    W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), "f32", seed=4)
    x, e = P.to_dev(X, "f32", torch), P.to_dev(E, "f32", torch)
    lib.set_kernel_variant(3)
    try:
        got = P.to_host(b.updat(x, e)); k32 = lib.last_kernel()
    finally:
        lib.set_kernel_variant(0)
    assert k32 == lib.K_UPDAT_STREAM
    P.assert_blocks(got, orc.updat(t, X.astype(np.float64), E.astype(np.float64), 1), "f32", b.blocks, "fp32 through the pieces")
    w = P.to_dev(W, "bf16", torch).requires_grad_()
    xb = P.to_dev(X, "bf16", torch)
    y = b(xb, w)
    y.backward(P.to_dev(E, "bf16", torch))
    ref = orc.updat_fast(t, P.to_host(xb).astype(np.float64), P.to_host(P.to_dev(E, "bf16", torch)).astype(np.float64), 1, dtype=np.float64)
    P.assert_blocks(P.to_host(w.grad), ref, "bf16", b.blocks, "w.grad")


def test_long_minibatches_take_the_big_window_plan(env):
    """BASELINE configs[3]'s layout (8192^2, 5 %): the host class holds two weight-gradient plans and picks per call -- 16 x 16-block windows (direct
    stores) below BlocksparseMatMul.LONG_MINIBATCH rows, 32 x 32-block windows (with direct blocks) from there on; both against the oracle."""
    torch, BSMM, lib = env
    lay = P.random_layout(256, 256, 0.05, 1234)
    b = BSMM(lay, block_size=32, feature_axis=1)
    tabs = b._tables_on(torch.device("cuda"))
    assert tabs.updat_plan_long is not None and int(tabs.updat_plan.host[2]) == 16 and int(tabs.updat_plan_long.host[2]) == 32
    t = orc.build_layout_luts(np.asarray(lay), 32)
    for N in (2048, 4096):
        W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), "bf16", seed=N)
        x, e = P.to_dev(X, "bf16", torch), P.to_dev(E, "bf16", torch)
        calls = []
        real = b._call_args
        b._call_args = lambda *a, **k: (calls.append(a[8]), real(*a, **k))[1]
        try:
            got = P.to_host(b.updat(x, e))
        finally:
            b._call_args = real
        assert lib.last_kernel() == lib.K_UPDAT_STREAM
        assert int(calls[-1].host[2]) == (32 if N >= b.LONG_MINIBATCH else 16), N        # window side of the plan the call ran with
        ws = sorted(set(range(0, b.blocks, 37)) | {b.blocks - 1})
        ref = orc.updat_blocks(t, P.to_host(x), P.to_host(e), 1, ws)
        P.assert_blocks(np.stack([got[i] for i in ws]), np.stack([ref[i] for i in ws]), "bf16", len(ws), ("sampled blocks", N))
    assert BSMM(P.random_layout(128, 128, 0.05, 1234), block_size=32, feature_axis=1)._tables_on(torch.device("cuda")).updat_plan_long is None     # 16 windows only
