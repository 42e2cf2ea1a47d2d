import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/scripts")
import numpy as np, torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib
from gpu_ref_bench_shapes import graph_us
from oracle import bsmm_oracle as O
lay = P.random_layout(64, 64, 0.2, 1234)
b = BlocksparseMatMul(lay, block_size=64, feature_axis=1)
t = O.build_layout_luts(lay, 64)
for N in (64, 200, 512, 2048):
    W, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), "bf16", seed=5)
    x, e = (P.to_dev(a, "bf16", torch) for a in (X, E))
    g = np.random.RandomState(1).uniform(-1, 2, b.blocks).astype(np.float32)
    dw = b.updat(x, e); k = _lib.last_kernel() & 255
    l2, _ = P.errors(P.to_host(dw), O.round_to(O.updat_fast(t, X, E, 1, np.float64), "bf16"))
    dwg = b.updat(x, e, gate=torch.from_numpy(g).cuda(), alpha=0.5)
    l2g, _ = P.errors(P.to_host(dwg), O.round_to(0.5 * O.updat_fast(t, X, E, 1, np.float64) * g[:, None, None], "bf16"))
    dwt = torch.empty(b.w_shape, dtype=torch.bfloat16, device="cuda")
    print("bs64 N %d: k%d l2 %.2e gated l2 %.2e updat %.1f us" % (N, k, l2, l2g, graph_us(lambda: b.updat(x, e, dw=dwt))), flush=True)
