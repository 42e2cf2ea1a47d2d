"""the reference's own bench layout (Barabasi-Albert + I, 4096^2, bsize 32) against a uniform layout with as many blocks: fprop / bprop / updat"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib

def timeit(fn, reps=100, warm=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

ba = P.ba_layout(128, 14, seed=1)
uni = P.random_layout(128, 128, ba.sum() / 128.0 / 128.0, 1234)
for name, lay in (("BA(128,14)+I", ba), ("uniform", uni)):
    b = BlocksparseMatMul(lay, block_size=32, feature_axis=1)
    g = torch.Generator(device="cuda").manual_seed(1)
    w = (torch.randn(b.w_shape, device="cuda", generator=g) * 0.05).bfloat16()
    x = (torch.randn(b.i_shape(8192), device="cuda", generator=g) * 0.1).bfloat16()
    dy = (torch.randn(b.o_shape(8192), device="cuda", generator=g) * 0.1).bfloat16()
    dw = torch.empty_like(w)
    rows = lay.sum(1); cols = lay.sum(0)
    print("%-13s blocks %d (max per row %d, per column %d): fprop %.1f bprop %.1f updat %.1f us" % (name, b.blocks, rows.max(), cols.max(),
          timeit(lambda: b.fprop(x, w)), timeit(lambda: b.bprop(dy, w)), timeit(lambda: b.updat(x, dy, dw=dw))), flush=True)
