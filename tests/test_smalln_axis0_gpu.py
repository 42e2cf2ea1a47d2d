"""Small minibatches on feature axis 0 (round 6: csrc/bsmm_xsmall0.h -- the regime of the reference's own benchmark, test/blocksparse_matmul_bench.py:37-78:
feature axis 0, minibatch 64) and the weight-gradient dispatch there: fprop / bprop / updat of the product API against the float64 oracle, the kernel
family asserted."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _parity as P
from oracle import bsmm_oracle as O


@pytest.fixture(scope="module")
def env():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import __graft_entry__ as g
    g.build()
    from blocksparse_amd import BlocksparseMatMul, _lib
    return torch, BlocksparseMatMul, _lib


LAYOUTS = [("dense 12x20", np.ones((12, 20), dtype=np.int32)), ("BA 64", P.ba_layout(64, 5, seed=1)), ("random 40x24", P.random_layout(40, 24, 0.3, seed=2)),
           ("empty rows and columns", np.eye(15, 40, dtype=np.int32)), ("single", np.ones((1, 1), dtype=np.int32)), ("one long column", np.ones((70, 1), dtype=np.int32))]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("bs", [32, 16, 8])
@pytest.mark.parametrize("name,layout", LAYOUTS)
def test_small_minibatch_xprop_axis0(env, name, layout, bs, dtype):
    torch, BSMM, lib = env
    b = BSMM(layout, block_size=bs, feature_axis=0)
    t = O.build_layout_luts(layout, bs)
    for N in (8, 64, 72, 200, 512):
        W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), dtype, seed=11 + N)
        w, x, e = (P.to_dev(a, dtype, torch) for a in (W, X, E))
        y = b.fprop(x, w)
        assert lib.last_kernel() & 255 == lib.K_XPROP_SMALL, (name, N, lib.last_kernel())
        dx = b.bprop(e, w)
        assert lib.last_kernel() & 255 == lib.K_XPROP_SMALL
        for what, got, ref in (("Y", y, O.fprop(t, X, W, 0)), ("DX", dx, O.bprop(t, E, W, 0))):
            l2, _ = P.errors(P.to_host(got), O.round_to(ref, dtype))
            assert l2 <= P.L2_BAR[dtype], (name, dtype, N, what, l2)
        # the same sums as the per-segment kernel up to the fp32 summation order
        lib.set_kernel_variant(2)
        try:
            y2 = b.fprop(x, w)
        finally:
            lib.set_kernel_variant(0)
        l2, _ = P.errors(P.to_host(y), P.to_host(y2))
        assert l2 <= P.L2_BAR[dtype], (name, N, l2)


@pytest.mark.gpu
@pytest.mark.parametrize("bs", [32, 16])
def test_small_minibatch_updat_axis0_takes_the_per_block_kernel(env, bs):
    """At the reference bench's shapes (hidden 2560 dense, Barabasi-Albert at 7680, minibatch 64) the weight gradient of feature axis 0 runs the
    per-block kernel where the windows are nearly empty or the minibatch is short (measured: scripts/gpu_a0_updat_sweep.py), against the oracle."""
    torch, BSMM, lib = env
    n = 2560 // bs
    for name, lay, N in (("dense", np.ones((n // 2, n // 2), dtype=np.int32), 64), ("BA", P.ba_layout(3 * n // 4, 6 if bs == 32 else 11, seed=1), 64)):
        b = BSMM(lay, block_size=bs, feature_axis=0)
        t = O.build_layout_luts(lay, bs)
        W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), "bf16", seed=5)
        x, e = (P.to_dev(a, "bf16", torch) for a in (X, E))
        dw = b.updat(x, e)
        k = lib.last_kernel() & 255
        if bs == 32:
            assert k == lib.K_UPDAT_BLOCK, (name, k)
        l2, _ = P.errors(P.to_host(dw), O.round_to(O.updat_fast(t, X, E, 0, np.float64), "bf16"))
        assert l2 <= P.L2_BAR["bf16"], (bs, name, l2)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("bs", [8, 16, 32])
def test_small_minibatch_updat_axis0(env, bs, dtype):
    """Feature axis 0 / short minibatches: bsize 8 one wave per PAIR of blocks (csrc/bsmm_updat.h::updat8_a0_pairs_kernel), bsize 16 / 32 one wave per
    block (updat_a0_wave_kernel) where the dispatch rules pick the per-block kernels -- odd block counts, ragged minibatches (N % 32 != 0), alpha /
    beta, a gate, two (x, dy) pairs -- against the float64 oracle.  (layouts without a plan-worthy window structure: the rules of bsize 16 may keep
    the windowed kernel; the kernel family is asserted for bsize 8 only, test_small_minibatch_updat_axis0_takes_the_per_block_kernel does the rest)"""
    torch, BSMM, lib = env
    for name, lay in (("BA 33", P.ba_layout(33, 3, seed=2)), ("dense 7x5", np.ones((7, 5), dtype=np.int32)), ("random 40x24", P.random_layout(40, 24, 0.3, seed=2))):
        b = BSMM(lay, block_size=bs, feature_axis=0)
        t = O.build_layout_luts(lay, bs)
        for N in (8, 64, 72, 200):
            W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), dtype, seed=3 + N)
            x, e, w0 = (P.to_dev(a, dtype, torch) for a in (X, E, W))
            dw = b.updat(x, e)
            assert bs != 8 or lib.last_kernel() & 255 == lib.K_UPDAT_BLOCK, (name, N, lib.last_kernel())
            l2, _ = P.errors(P.to_host(dw), O.round_to(O.updat(t, X, E, 0), dtype))
            assert l2 <= P.L2_BAR[dtype], (name, dtype, N, l2)
            dw2 = b.updat(x, e, alpha=0.5, beta=2.0, dw=w0.clone())
            ref = 0.5 * O.updat(t, X, E, 0) + 2.0 * np.asarray(P.to_host(w0), dtype=np.float64)
            l2, _ = P.errors(P.to_host(dw2), O.round_to(ref, dtype))
            assert l2 <= P.L2_BAR[dtype], (name, dtype, N, "alpha/beta", l2)
        N = 64
        W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), dtype, seed=9)
        _, X2, E2 = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), dtype, seed=10)
        xs = [P.to_dev(a, dtype, torch) for a in (X, X2)]
        es = [P.to_dev(a, dtype, torch) for a in (E, E2)]
        g = np.random.RandomState(4).uniform(-1, 2, b.blocks).astype(np.float32)
        dw = b.updat(xs, es, gate=torch.from_numpy(g).cuda())
        assert bs != 8 or lib.last_kernel() & 255 == lib.K_UPDAT_BLOCK
        ref = (O.updat(t, X, E, 0) + O.updat(t, X2, E2, 0)) * g[:, None, None]
        l2, _ = P.errors(P.to_host(dw), O.round_to(ref, dtype))
        assert l2 <= P.L2_BAR[dtype], (name, dtype, "pairs + gate", l2)


@pytest.mark.gpu
def test_small_minibatch_updat_bsize64(env):
    """bsize 64 (feature axis 1: the reference's other block size there) at short minibatches: the one-wave-per-block kernel writes every quadrant
    into its 64 x 64 block (round 6; before: the streaming kernel, 35 us at N = 64 against 5) -- plain, gated with alpha, beta accumulate -- vs the oracle."""
    torch, BSMM, lib = env
    lay = P.random_layout(20, 12, 0.3, seed=6)
    b = BSMM(lay, block_size=64, feature_axis=1)
    t = O.build_layout_luts(lay, 64)
    for N in (64, 200):
        W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), "bf16", seed=5 + N)
        x, e, w0 = (P.to_dev(a, "bf16", torch) for a in (X, E, W))
        ref = O.updat(t, X, E, 1)
        dw = b.updat(x, e)
        assert lib.last_kernel() & 255 == lib.K_UPDAT_BLOCK_TR, lib.last_kernel()
        l2, _ = P.errors(P.to_host(dw), O.round_to(ref, "bf16"))
        assert l2 <= P.L2_BAR["bf16"], (N, l2)
        g = np.random.RandomState(1).uniform(-1, 2, b.blocks).astype(np.float32)
        dwg = b.updat(x, e, gate=torch.from_numpy(g).cuda(), alpha=0.5, beta=2.0, dw=w0.clone())
        assert lib.last_kernel() & 255 == lib.K_UPDAT_BLOCK_TR
        refg = 0.5 * ref * g[:, None, None] + 2.0 * np.asarray(P.to_host(w0), dtype=np.float64)
        l2, _ = P.errors(P.to_host(dwg), O.round_to(refg, "bf16"))
        assert l2 <= P.L2_BAR["bf16"], (N, "gated", l2)


@pytest.mark.gpu
def test_small_minibatch_axis0_fuzz(env):
    """Eighteen seeded random cases over the small-minibatch kernels of feature axis 0: layouts of 1 .. 60 blocks a side at 3 .. 100 %, block sizes
    8 / 16 / 32, minibatches that are multiples of 8 up to 1024, bf16 / fp16 -- all three passes against the float64 oracle (the kernel family of
    fprop asserted where the rule of bsmm_api.hip must take the small kernel: N <= 512)."""
    torch, BSMM, lib = env
    rng = np.random.default_rng(20260930)
    for it in range(18):
        bs = (32, 16, 8)[it % 3]
        CB, KB = int(rng.integers(1, 61)), int(rng.integers(1, 61))
        dens = float(rng.choice([0.03, 0.1, 0.3, 0.6, 1.0]))
        N = 8 * int(rng.integers(1, 129))
        dt = ("bf16", "f16")[(it // 3) & 1]
        lay = P.random_layout(CB, KB, dens, seed=500 + it)
        if not lay.any():
            lay[0, 0] = 1
        b = BSMM(lay, block_size=bs, feature_axis=0)
        t = O.build_layout_luts(lay, bs)
        W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), dt, seed=900 + it)
        w, x, e = (P.to_dev(a, dt, torch) for a in (W, X, E))
        y = b.fprop(x, w)
        if N <= 512:
            assert lib.last_kernel() & 255 == lib.K_XPROP_SMALL, (it, bs, CB, KB, dens, N, lib.last_kernel())
        for what, got, ref in (("Y", y, O.fprop(t, X, W, 0)), ("DX", b.bprop(e, w), O.bprop(t, E, W, 0)), ("DW", b.updat(x, e), O.updat(t, X, E, 0))):
            l2, _ = P.errors(P.to_host(got), O.round_to(ref, dt))
            assert l2 <= P.L2_BAR[dt], (it, bs, CB, KB, dens, N, dt, what, l2)


@pytest.mark.gpu
def test_gated_images_under_stream_capture(env):
    """A gated call on the weight-image path inside a hipGraph capture: no host sync (an unseen gate counts as "general" there: two images), the
    replayed result equals the eager one."""
    torch, BSMM, lib = env
    lay = P.random_layout(40, 24, 0.3, seed=21)
    b = BSMM(lay, block_size=32, feature_axis=1)
    N = 1024
    W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), "bf16", seed=3)
    w, x = P.to_dev(W, "bf16", torch), P.to_dev(X, "bf16", torch)
    g = torch.from_numpy((np.random.RandomState(2).rand(b.blocks) < 0.7).astype(np.float32)).cuda()
    b.gate_kind = "general"
    want = b.fprop(x, w, gate=g)                       # (eager, two images: what the capture will take for a gate it has not looked at)
    b.gate_kind = "auto"
    b._gate_kind_hit = None
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        b.fprop(x, w)                                  # warm the ungated tables / plans outside the capture
    torch.cuda.current_stream().wait_stream(side)
    b._doubled()._tables_on(x.device)                  # (plans are host work: built before the capture)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        got = b.fprop(x, w, gate=g)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("bs", [16, 8])
@pytest.mark.parametrize("name,layout", LAYOUTS)
def test_small_minibatch_xprop_axis1_narrow_blocks(env, name, layout, bs, dtype):
    """bsize 16 / 8 on feature axis 1 at short minibatches (csrc/bsmm_xsmall.h::xsmall_narrow_kernel, round 6): fprop / bprop against the float64
    oracle, any minibatch (ragged row tiles, N = 1), odd entry counts (bsize 8 multiplies PAIRS of entries)."""
    torch, BSMM, lib = env
    b = BSMM(layout, block_size=bs, feature_axis=1)
    t = O.build_layout_luts(layout, bs)
    for N in (1, 50, 64, 200, 512):
        W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), dtype, seed=17 + N)
        w, x, e = (P.to_dev(a, dtype, torch) for a in (W, X, E))
        y = b.fprop(x, w)
        assert (lib.last_kernel() & 255 == lib.K_XPROP_SMALL) == (N <= (512 if bs == 8 else 256)), (name, N, lib.last_kernel())
        dx = b.bprop(e, w)
        assert (lib.last_kernel() & 255 == lib.K_XPROP_SMALL) == (N <= (512 if bs == 8 else 64)), (name, N, lib.last_kernel())
        for what, got, ref in (("Y", y, O.fprop(t, X, W, 1)), ("DX", dx, O.bprop(t, E, W, 1))):
            l2, _ = P.errors(P.to_host(got), O.round_to(ref, dtype))
            assert l2 <= P.L2_BAR[dtype], (name, bs, dtype, N, what, l2)
