import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib
td = torch.bfloat16
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for bs, dens in ((16, 0.1), (8, 0.1)):
    CB = 4096 // bs
    for axis in (0, 1):
        b = BlocksparseMatMul(P.random_layout(CB, CB, dens, seed=1234), block_size=bs, feature_axis=axis)
        for N in (8192,):
            w = (torch.randn(b.w_shape, device="cuda") * 0.01).to(td)
            x = (torch.randn(b.i_shape(N), device="cuda") * 0.1).to(td)
            dy = (torch.randn(b.o_shape(N), device="cuda") * 0.1).to(td)
            fl = 2.0 * b.blocks * bs * bs * N
            tf, tb, tu = timeit(lambda: b.fprop(x, w)), timeit(lambda: b.bprop(dy, w)), timeit(lambda: b.updat(x, dy))
            print("bs%d a%d d%.2f N%-5d fprop %.3f ms %6.1f TF | bprop %.3f ms %6.1f TF | updat %.3f ms %6.1f TF" % (bs, axis, dens, N, tf, fl/tf/1e9, tb, fl/tb/1e9, tu, fl/tu/1e9), flush=True)
