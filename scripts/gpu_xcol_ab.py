"""bf16 bs32 xprop/updat timing at the bench workload (quick A/B of compile-time variants)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul
def timeit(fn, reps=50):
    for _ in range(15): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
axis = int(sys.argv[1]) if len(sys.argv) > 1 else 1
import time
_x = torch.randn(8192, 8192, device="cuda"); _t = time.time()
while time.time() - _t < 0.7: _x @ _x          # boost clock first
for d in (0.1, 0.2, 0.5):
    b = BlocksparseMatMul(P.random_layout(128, 128, d, seed=1234), block_size=32, feature_axis=axis)
    N = 8192
    w = (torch.randn(b.w_shape, device="cuda") * 0.01).bfloat16()
    x = (torch.randn(b.i_shape(N), device="cuda") * 0.1).bfloat16()
    dy = (torch.randn(b.o_shape(N), device="cuda") * 0.1).bfloat16()
    fl = 2.0 * b.blocks * 1024 * N
    tf, tb, tu = timeit(lambda: b.fprop(x, w)), timeit(lambda: b.bprop(dy, w)), timeit(lambda: b.updat(x, dy))
    print("%s a%d d%.2f fprop %.3f ms %6.1f TF | bprop %.3f ms %6.1f TF | updat %.3f ms %6.1f TF" % (os.environ.get("TAG", ""), axis, d, tf, fl/tf/1e9, tb, fl/tb/1e9, tu, fl/tu/1e9), flush=True)

if os.environ.get("NSWEEP"):
    b = BlocksparseMatMul(P.random_layout(128, 128, 0.2, seed=1234), block_size=32, feature_axis=axis)
    w = (torch.randn(b.w_shape, device="cuda") * 0.01).bfloat16()
    for N in (512, 1024, 2048, 3072, 4096, 6144):
        x = (torch.randn(b.i_shape(N), device="cuda") * 0.1).bfloat16()
        dy = (torch.randn(b.o_shape(N), device="cuda") * 0.1).bfloat16()
        tf, tb = timeit(lambda: b.fprop(x, w)), timeit(lambda: b.bprop(dy, w))
        print("%s a%d N=%d fprop %.1f us | bprop %.1f us" % (os.environ.get("TAG", ""), axis, N, tf * 1e3, tb * 1e3), flush=True)
