"""bs16 (BASELINE configs[2]) xprop/updat timing for A/B of compile-time variants."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul
def timeit(fn, reps=100):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for axis in (0, 1):
    b = BlocksparseMatMul(P.random_layout(256, 256, 0.1, seed=1234), block_size=16, feature_axis=axis)
    N = 8192
    w = (torch.randn(b.w_shape, device="cuda") * 0.01).bfloat16()
    x = (torch.randn(b.i_shape(N), device="cuda") * 0.1).bfloat16()
    dy = (torch.randn(b.o_shape(N), device="cuda") * 0.1).bfloat16()
    fl = 2.0 * b.blocks * 256 * N
    tf, tb, tu = timeit(lambda: b.fprop(x, w)), timeit(lambda: b.bprop(dy, w)), timeit(lambda: b.updat(x, dy))
    print("%s bs16 a%d fprop %.3f ms %6.1f TF | bprop %.3f ms %6.1f TF | updat %.3f ms %6.1f TF" % (
        os.environ.get("TAG", ""), axis, tf, fl/tf/1e9, tb, fl/tb/1e9, tu, fl/tu/1e9), flush=True)
