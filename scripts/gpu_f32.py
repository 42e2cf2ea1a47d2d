"""fp32 bsize-32 xprop: grouped kernel (xcol32f) vs the per-segment kernel (variant 2), both axes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib
L = _lib.load()
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for axis in (1, 0):
    b = BlocksparseMatMul(P.random_layout(128, 128, 0.2, seed=1234), block_size=32, feature_axis=axis)
    for N in (2048, 8192):
        w = torch.randn(b.w_shape, device="cuda") * 0.01
        x = torch.randn(b.i_shape(N), device="cuda") * 0.1
        dy = torch.randn(b.o_shape(N), device="cuda") * 0.1
        fl = 2.0 * b.blocks * 1024 * N
        res = {}
        for v in (0, 2):
            L.bsmm_set_kernel_variant(v)
            y = b.fprop(x, w); dx = b.bprop(dy, w)
            tf, tb, tu = timeit(lambda: b.fprop(x, w)), timeit(lambda: b.bprop(dy, w)), timeit(lambda: b.updat(x, dy))
            res[v] = (y, dx)
            print("f32 a%d N%-5d variant %d fprop %.3f ms %6.1f TF | bprop %.3f ms %6.1f TF | updat %.3f ms %6.1f TF" % (axis, N, v, tf, fl/tf/1e9, tb, fl/tb/1e9, tu, fl/tu/1e9), flush=True)
        L.bsmm_set_kernel_variant(0)
        for nm, p, q in zip(("Y", "DX"), res[0], res[2]):
            print("   %s plan-vs-generic rel l2 %.2e" % (nm, ((p - q).double().norm() / q.double().norm()).item()))
