"""randomised layouts whose 16 x 16-block windows overflow (bsize 32, feature axis 1): the streaming weight-gradient kernel with DIRECT blocks against the
float64 oracle, every block (tests/_parity.py::assert_blocks), random minibatch (ragged too), 1 .. 3 pairs, alpha / beta, gate.  argv: seed, cases"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import _parity as P
from oracle import bsmm_oracle as orc
from blocksparse_amd import BlocksparseMatMul, _lib
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
_lib.set_kernel_variant(3)
with_direct = 0
for it in range(ncase):
    CB, KB = int(rng.integers(10, 131)), int(rng.integers(10, 131))
    dens = float(rng.uniform(0.14, 0.30))
    lay = (rng.random((CB, KB)) < dens).astype(np.int32)
    for _ in range(int(rng.integers(0, 3))):                      # a crowded window or two
        r0, c0 = 16 * int(rng.integers(0, (CB + 15) // 16)), 16 * int(rng.integers(0, (KB + 15) // 16))
        sub = lay[r0:r0 + 16, c0:c0 + 16]
        sub |= (rng.random(sub.shape) < 0.12).astype(np.int32)
    lay[0, 0] = 1
    dtype = str(rng.choice(["bf16", "f16"]))
    N = int(rng.choice([8, 24, 40, 136, 520, 1000, 2048, 3333, 8192]))
    npair = int(rng.integers(1, 4))
    b = BlocksparseMatMul(lay, block_size=32, feature_axis=1)
    hp = b._tables_on(torch.device("cuda")).updat_plan.host
    ndir = int(hp[28]) if int(hp[0]) == 0x42535532 else -1
    with_direct += ndir > 0
    t = orc.build_layout_luts(lay, 32)
    xs, es, ref = [], [], 0.0
    for p in range(npair):
        W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), dtype, seed=100 * it + p)
        xs.append(P.to_dev(X, dtype, torch)); es.append(P.to_dev(E, dtype, torch))
        ref = ref + orc.updat_fast(t, P.to_host(xs[-1]).astype(np.float64), P.to_host(es[-1]).astype(np.float64), 1, dtype=np.float64)
    alpha, beta = (1.0, 0.0) if rng.random() < 0.5 else (float(rng.uniform(0.25, 2)), float(rng.uniform(-1, 1)))
    dw0 = P.to_dev(rng.standard_normal(b.w_shape).astype(np.float32) * 0.05, dtype, torch)
    g = None
    if rng.random() < 0.4:
        g = rng.random(b.blocks).astype(np.float32) * 2 - 0.5
        g[rng.random(b.blocks) < 0.2] = 0
    got = P.to_host(b.updat(xs, es, alpha=alpha, beta=beta, dw=dw0.clone(), gate=None if g is None else torch.from_numpy(g).cuda()))
    k = _lib.last_kernel()
    want = alpha * ref * (1.0 if g is None else g.astype(np.float64)[:, None, None]) + beta * P.to_host(dw0).astype(np.float64)
    try:
        rep = P.assert_blocks(got, want, dtype, b.blocks, it)
        flag = ""
    except AssertionError as ex:
        rep = P.block_report(got, want, dtype, b.blocks); flag = "   <<<<<< FAIL"
    print("%3dx%3d d%.2f %s N%4d pairs %d a%.2f b%+.2f gate%d items %d direct %2d k%d  tensor %.1e block %.1e bad %d%s" % (
        CB, KB, lay.mean(), dtype, N, npair, alpha, beta, g is not None, int(hp[4]), ndir, k, rep["tensor_l2"], rep["block_l2"], rep["elem_bad"], flag), flush=True)
_lib.set_kernel_variant(0)
print("cases", ncase, "with direct blocks", with_direct)
