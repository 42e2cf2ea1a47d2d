"""GPU parity of the row-owner weight-gradient kernel (bsize 16, feature axis 0, 16-bit types: csrc/bsmm_updat16_rows.h, the 'BSU6' section of
the bsize-16 'BSUP' plan, round 5) against the float64 oracle (oracle/bsmm_oracle.py::updat, restating blocksparse/matmul.py:401-419 with the
kernel semantics of alpha / beta / pairs, src/blocksparse_matmul_op_gpu.cu:2684-2814), every block of DW, with the kernel family asserted
through bsmm_args.trace.  The windowed kernel (plan option PLAN_UPDAT16_WINDOWED) runs beside it: same inputs, same bar."""
import os

import numpy as np
import pytest

import _parity as P
from oracle import bsmm_oracle as orc

pytestmark = pytest.mark.gpu

# Every random draw of this file is seeded; scripts/gpu_seed_sweep.py re-runs the three configs[2] / random-shape / fp32 tests under many values
# of BSMM_TEST_SEED on one lease (the criterion must hold for every seed, not for the lucky one).
SEED = int(os.environ.get("BSMM_TEST_SEED", "0"))


@pytest.fixture(scope="module")
def env():
    import torch
    from blocksparse_amd import BlocksparseMatMul, _lib
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    _lib.load()
    return torch, BlocksparseMatMul, _lib


CASES = [
    # name, layout, N, dtype, split (0 = the library's), forced (call variant 3: the plan kernel whatever the size heuristic says)
    ("one window, dense 4x4", np.ones((4, 4), dtype=np.int32), 512, "bf16", 0, True),
    ("40x24 30 %: 16-column windows, ragged last chunk", P.random_layout(40, 24, 0.3, seed=2), 1000, "bf16", 0, True),
    ("33x35 25 %: grid not a multiple of the window", P.random_layout(33, 35, 0.25, seed=3), 520, "f16", 0, True),
    ("70x96 10 %: 32-column windows, four parts", P.random_layout(70, 96, 0.10, seed=4), 2048, "bf16", 4, False),
    ("70x96 10 %: one part (direct store)", P.random_layout(70, 96, 0.10, seed=4), 2048, "f16", 1, False),
    ("64x64 10 %: one chunk of 8 entries", P.random_layout(64, 64, 0.10, seed=5), 8, "bf16", 1, False),
    ("64x64 10 %: more parts than chunks", P.random_layout(64, 64, 0.10, seed=5), 192, "bf16", 8, False),
    ("256x256 10 % (BASELINE configs[2]'s layout), library's split", P.random_layout(256, 256, 0.10, seed=1234), 4096, "bf16", 0, False),
    ("256x256 20 %: 16-column windows, library's split", P.random_layout(256, 256, 0.20, seed=1234), 2048, "f16", 0, False),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_row_owner_updat_against_the_oracle(env, case):
    torch, BSMM, lib = env
    name, lay, N, dt, split, forced = case
    b = BSMM(lay, block_size=16, feature_axis=0, updat_split=split)
    bw = BSMM(lay, block_size=16, feature_axis=0, plan_options=lib.PLAN_UPDAT16_WINDOWED)
    assert int(b._tables_on(torch.device("cuda")).updat_plan.host[8]) > 0 and int(bw._tables_on(torch.device("cuda")).updat_plan.host[8]) == 0
    W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), dt, seed=11)
    x, e = P.to_dev(X, dt, torch), P.to_dev(E, dt, torch)
    dw0 = P.to_dev(np.random.default_rng(3).standard_normal(b.w_shape).astype(np.float32) * 0.05, dt, torch)
    lib.set_kernel_variant(3 if forced else 0)
    try:
        d1 = P.to_host(b.updat(x, e)); k1 = lib.last_kernel()
        d2 = P.to_host(b.updat(x, e, alpha=0.5, beta=2.0, dw=dw0.clone())); k2 = lib.last_kernel()
        d3 = P.to_host(b.updat([x, x], [e, e])); k3 = lib.last_kernel()
        d1b = P.to_host(b.updat(x, e))
        dw_ = P.to_host(bw.updat(x, e)); kw = lib.last_kernel()
    finally:
        lib.set_kernel_variant(0)
    assert (k1, k2, k3) == (lib.K_UPDAT16_ROWS,) * 3 and kw != lib.K_UPDAT16_ROWS, (k1, k2, k3, kw)
    t = orc.build_layout_luts(np.asarray(lay), 16)
    Xh, Eh = P.to_host(x).astype(np.float64), P.to_host(e).astype(np.float64)
    ref = orc.updat(t, Xh, Eh, 0)
    for got, want, what in ((d1, ref, "dw"), (d2, 0.5 * ref + 2.0 * P.to_host(dw0).astype(np.float64), "alpha / beta"), (d3, 2.0 * ref, "two pairs"),
                            (dw_, ref, "windowed kernel")):
        P.assert_blocks(got, want, dt, b.blocks, (name, what))
    assert np.array_equal(d1, d1b)                       # no atomics: the parts are added in a fixed order


def test_row_owner_updat_falls_back_where_it_cannot_run(env):
    """N % 8 != 0 (row pieces are not 16-byte aligned), a gated call, fp32 and minibatches too small to pay take other kernels -- with the
    same results."""
    torch, BSMM, lib = env
    lay = P.random_layout(70, 96, 0.10, seed=4)
    b = BSMM(lay, block_size=16, feature_axis=0)
    t = orc.build_layout_luts(np.asarray(lay), 16)
    for N, dt, gated in ((1004, "bf16", False), (2048, "bf16", True), (2048, "f32", False), (128, "bf16", False)):
        W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), dt, seed=5)
        x, e = P.to_dev(X, dt, torch), P.to_dev(E, dt, torch)
        gate = torch.rand(b.blocks, device="cuda", generator=torch.Generator(device="cuda").manual_seed(N)) if gated else None
        got = P.to_host(b.updat(x, e, gate=gate))
        assert lib.last_kernel() != lib.K_UPDAT16_ROWS, (N, dt, gated, lib.last_kernel())
        ref = orc.updat(t, P.to_host(x).astype(np.float64), P.to_host(e).astype(np.float64), 0, gate=None if gate is None else gate.cpu().numpy())
        l2, _ = P.errors(got, orc.round_to(ref, dt))
        assert l2 <= P.L2_BAR[dt], (N, dt, gated, l2)


def test_row_owner_updat_at_configs2(env):
    """BASELINE configs[2] itself (4096^2, bsize 16, 10 %, feature axis 0, bf16, N = 8192), production dispatch: the row-owner kernel with the
    library's split, every block against the float64 oracle."""
    torch, BSMM, lib = env
    lay = P.random_layout(256, 256, 0.10, seed=1234)
    b = BSMM(lay, block_size=16, feature_axis=0)
    N = 8192
    g = torch.Generator(device="cuda").manual_seed(SEED + 9)
    x = (torch.randn(b.i_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
    e = (torch.randn(b.o_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
    dw = P.to_host(b.updat(x, e))
    assert lib.last_kernel() == lib.K_UPDAT16_ROWS
    t = orc.build_layout_luts(np.asarray(lay), 16)
    ref = orc.updat_fast(t, P.to_host(x), P.to_host(e), 0, dtype=np.float64)
    P.assert_blocks(dw, ref, "bf16", b.blocks, "configs[2] dw")
    # a gated call stays on the kernel: its finalize pass scales the sums of block w by gate[w] before the one rounding
    gate = torch.rand(b.blocks, device="cuda", generator=torch.Generator(device="cuda").manual_seed(SEED + 10)) * 2 - 0.5
    gate[::97] = 0.0                                     # gate == 0 blocks come out exactly zero
    dwg = P.to_host(b.updat(x, e, gate=gate))
    assert lib.last_kernel() == lib.K_UPDAT16_ROWS
    P.assert_blocks(dwg, ref * gate.cpu().numpy().astype(np.float64)[:, None, None], "bf16", b.blocks, "configs[2] gated dw")


def test_row_owner_updat_random_shapes(env):
    """Sixteen random layouts (20 .. 150 blocks a side, 3 .. 25 %), minibatches that are multiples of 8 up to 1500, 1 .. 8 parts, bf16 / fp16,
    alpha / beta: whatever the section builder makes of them (32- or 16-column windows; none for the densest: those runs must take another
    kernel), every block against the float64 oracle."""
    torch, BSMM, lib = env
    rng = np.random.default_rng(77 + SEED)
    ran = 0
    for it in range(16):
        CB, KB = int(rng.integers(20, 151)), int(rng.integers(20, 151))
        dens = float(rng.uniform(0.03, 0.25))
        N = 8 * int(rng.integers(1, 188))
        split = int(rng.integers(1, 9))
        dt = ("bf16", "f16")[it & 1]
        lay = P.random_layout(CB, KB, dens, seed=100 + it)
        b = BSMM(lay, block_size=16, feature_axis=0, updat_split=split)
        has = int(b._tables_on(torch.device("cuda")).updat_plan.host[8]) > 0
        W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), dt, seed=it + 1000 * SEED)
        x, e = P.to_dev(X, dt, torch), P.to_dev(E, dt, torch)
        dw0 = P.to_dev(rng.standard_normal(b.w_shape).astype(np.float32) * 0.05, dt, torch)
        got = P.to_host(b.updat(x, e, alpha=1.5, beta=-0.5, dw=dw0.clone()))
        assert (lib.last_kernel() == lib.K_UPDAT16_ROWS) == has, (it, CB, KB, dens, N, split, lib.last_kernel())
        ran += has
        t = orc.build_layout_luts(np.asarray(lay), 16)
        ref = 1.5 * orc.updat(t, P.to_host(x).astype(np.float64), P.to_host(e).astype(np.float64), 0) - 0.5 * P.to_host(dw0).astype(np.float64)
        P.assert_blocks(got, ref, dt, b.blocks, (it, CB, KB, dens, N, split, dt))
    assert ran >= (10 if SEED == 0 else 4)


def test_fp32_updat_through_the_row_owner_kernel(env):
    """fp32 weight gradient of bsize 16 on feature axis 0 (the reference's primary fp32 layout, src/blocksparse_matmul_op_gpu.cu:1395-1835):
    X and DY split into three bf16 pieces each, the six significant piece products as six pairs of ONE launch of the row-owner kernel, fp32
    finalize with alpha / beta / gate -- against the float64 oracle at the fp32 bar; an Inf / NaN input raises the split's flag and the
    per-block fp32 kernel computes the call (same NaN / Inf sets as the oracle)."""
    torch, BSMM, lib = env
    lay = P.random_layout(256, 256, 0.10, seed=1234)
    b = BSMM(lay, block_size=16, feature_axis=0)
    N = 2048
    t = orc.build_layout_luts(np.asarray(lay), 16)
    W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), "f32", seed=3 + SEED)
    E = (E * 1e-3).astype(np.float32)
    X[5, 7] = np.float32(3.4e38)                               # finite, beyond the bf16 range: the split stays exact
    x, e = P.to_dev(X, "f32", torch), P.to_dev(E, "f32", torch)
    gate = torch.rand(b.blocks, device="cuda", generator=torch.Generator(device="cuda").manual_seed(SEED + 5))
    dw0 = (np.random.default_rng(5).standard_normal(b.w_shape) * 0.1).astype(np.float32)
    got = P.to_host(b.updat(x, e, alpha=0.5, beta=2.0, dw=P.to_dev(dw0.copy(), "f32", torch), gate=gate))
    assert lib.last_kernel() == lib.K_UPDAT16_ROWS
    ref = orc.updat(t, X.astype(np.float64), E.astype(np.float64), 0, alpha=0.5, beta=2.0, dw_in=dw0, gate=gate.cpu().numpy())
    P.assert_blocks(got, ref, "f32", b.blocks, "fp32 through the row-owner kernel")
    X[9, 100] = np.inf
    E[300, 11] = np.nan
    got2 = P.to_host(b.updat(P.to_dev(X, "f32", torch), P.to_dev(E, "f32", torch)))
    with np.errstate(invalid="ignore", over="ignore"):
        ref2 = orc.updat(t, X.astype(np.float64), E.astype(np.float64), 0)
    assert np.isnan(ref2).any() and np.isinf(ref2).any()
    assert np.array_equal(np.isnan(got2), np.isnan(ref2))
    inf = np.isinf(ref2)
    assert np.array_equal(np.isinf(got2), inf) and np.array_equal(np.sign(got2[inf]), np.sign(ref2[inf]))
