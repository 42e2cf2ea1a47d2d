"""randomised layouts at bsize 64 (feature axis 1, through bsmm_args.bsize = 64) and bsize 8 (both axes, super-block plans) against the float64 oracle"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import _parity as P
from oracle import bsmm_oracle as orc
from blocksparse_amd import BlocksparseMatMul, _lib
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
worst = 0.0
_lib.set_kernel_variant(3)
for it in range(ncase):
    bs = int(rng.choice([64, 8]))
    if bs == 64:
        CB, KB, axis = int(rng.integers(1, 24)), int(rng.integers(1, 24)), 1
    else:
        CB, KB, axis = 4 * int(rng.integers(1, 30)), 4 * int(rng.integers(1, 30)), int(rng.integers(0, 2))
    dens = float(rng.choice([0.05, 0.1, 0.3, 1.0]))
    lay = rng.random((CB, KB)) < dens
    lay[rng.integers(0, CB), rng.integers(0, KB)] = True
    dtype = str(rng.choice(["bf16", "f16"]))
    N = int(rng.choice([8, 40, 128, 264, 520, 1032]))
    b = BlocksparseMatMul(lay, block_size=bs, feature_axis=axis)
    t = orc.build_layout_luts(lay, bs)
    W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), dtype, seed=it)
    w, x, e = P.to_dev(W, dtype, torch), P.to_dev(X, dtype, torch), P.to_dev(E, dtype, torch)
    g = None
    if bs == 64 and rng.random() < 0.4:
        g = rng.random(b.blocks).astype(np.float32) * 2 - 0.5
        g[rng.random(b.blocks) < 0.3] = 0; g[rng.random(b.blocks) < 0.3] = 1
    tg = torch.from_numpy(g).cuda() if g is not None else None
    y = P.to_host(b.fprop(x, w, gate=tg)); kx = _lib.last_kernel()
    dx = P.to_host(b.bprop(e, w, gate=tg))
    dw0 = orc.round_to(rng.normal(size=b.w_shape).astype(np.float32) * 0.05, dtype)
    dw = P.to_host(b.updat(x, e, alpha=0.5, beta=2.0, dw=P.to_dev(dw0, dtype, torch))); ku = _lib.last_kernel()
    errs = [P.errors(y, orc.round_to(orc.fprop(t, X, W, axis, gate=g), dtype))[0], P.errors(dx, orc.round_to(orc.bprop(t, E, W, axis, gate=g), dtype))[0],
            P.errors(dw, orc.round_to(orc.updat(t, X, E, axis, alpha=0.5, beta=2.0, dw_in=dw0), dtype))[0]]
    worst = max(worst, max(errs))
    print("bs%d %dx%d d%.2f a%d %s N%4d gate%d k%d/%d  %.1e %.1e %.1e" % (bs, CB, KB, dens, axis, dtype, N, g is not None, kx, ku, *errs), flush=True)
    assert max(errs) <= 1e-3, "FAIL"
print("worst", worst)
