"""randomised shapes: forced plan kernels (bsize 32 / 16, both axes, bf16 / f16) against the float64 oracle"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import _parity as P
from oracle import bsmm_oracle as orc
from blocksparse_amd import BlocksparseMatMul, _lib
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 60
worst = 0.0
_lib.set_kernel_variant(3)
for it in range(ncase):
    bs = int(rng.choice([32, 32, 16]))
    big = os.environ.get("FUZZ_BIG") == "1"
    CB, KB = int(rng.integers(1, 200 if big else 70)), int(rng.integers(1, 200 if big else 70))
    dens = float(rng.choice([0.03, 0.1, 0.2, 0.5, 1.0]))
    lay = rng.random((CB, KB)) < dens
    lay[rng.integers(0, CB), rng.integers(0, KB)] = True
    axis = int(rng.integers(0, 2))
    dtype = str(rng.choice(["bf16", "f16"]))
    N = int(rng.choice([8, 40, 128, 200, 264, 520] + ([1032, 2056, 4096] if os.environ.get("FUZZ_BIG") == "1" else [])))
    if axis == 0: N = (N + 7) // 8 * 8
    b = BlocksparseMatMul(lay, block_size=bs, feature_axis=axis)
    t = orc.build_layout_luts(lay, bs)
    W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), dtype, seed=it)
    w, x, e = P.to_dev(W, dtype, torch), P.to_dev(X, dtype, torch), P.to_dev(E, dtype, torch)
    use_gate = bs == 32 and rng.random() < 0.3
    g = (rng.random(b.blocks).astype(np.float32) * 2 - 0.5) if use_gate else None
    if g is not None: g[rng.random(b.blocks) < 0.3] = 0; g[rng.random(b.blocks) < 0.3] = 1
    tg = torch.from_numpy(g).cuda() if g is not None else None
    y = P.to_host(b.fprop(x, w, gate=tg)); kf = _lib.last_kernel()
    dx = P.to_host(b.bprop(e, w, gate=tg))
    dw = P.to_host(b.updat(x, e)); ku = _lib.last_kernel()
    errs = (P.errors(y, orc.round_to(orc.fprop(t, X, W, axis, gate=g), dtype))[0], P.errors(dx, orc.round_to(orc.bprop(t, E, W, axis, gate=g), dtype))[0],
            P.errors(dw, orc.round_to(orc.updat(t, X, E, axis), dtype))[0])
    worst = max(worst, *errs)
    flag = "" if max(errs) <= P.L2_BAR[dtype] else "   <<<<<< FAIL"
    print("bs%d %2dx%2d d%.2f a%d %s N%3d gate%d k%d/%d  %.1e %.1e %.1e%s" % (bs, CB, KB, dens, axis, dtype, N, use_gate, kf, ku, *errs, flag), flush=True)
_lib.set_kernel_variant(0)
print("worst", worst)
