"""BASELINE configs[3] per-GPU shard (8192^2, bs 32, 5 %, N = 512 and 4096 (= one GPU)): plan kernels (variant 0) vs generic (2)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib
L = _lib.load()
def timeit(fn, reps=100):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
_x = torch.randn(8192, 8192, device="cuda"); _t = time.time()
while time.time() - _t < 0.7: _x @ _x
b = BlocksparseMatMul(P.random_layout(256, 256, 0.05, seed=1234), block_size=32, feature_axis=1)
for N in (512, 1024, 4096):
    w = (torch.randn(b.w_shape, device="cuda") * 0.01).bfloat16()
    x = (torch.randn(b.i_shape(N), device="cuda") * 0.1).bfloat16()
    dy = (torch.randn(b.o_shape(N), device="cuda") * 0.1).bfloat16()
    fl = 2.0 * b.blocks * 1024 * N
    for v in (0, 2, 3):
        L.bsmm_set_kernel_variant(v)
        tf, tb, tu = timeit(lambda: b.fprop(x, w)), timeit(lambda: b.bprop(dy, w)), timeit(lambda: b.updat(x, dy))
        print("cfg3 N=%d variant %d: fprop %.1f us %5.0f TF | bprop %.1f us %5.0f TF | updat %.1f us %5.0f TF" % (N, v, tf*1e3, fl/tf/1e9, tb*1e3, fl/tb/1e9, tu*1e3, fl/tu/1e9), flush=True)
    L.bsmm_set_kernel_variant(0)
# headline layout, minibatch sweep (variant 0 only): the cost model must not make these worse
b = BlocksparseMatMul(P.random_layout(128, 128, 0.2, seed=1234), block_size=32, feature_axis=1)
for N in (512, 1024, 2048, 4096, 8192):
    w = (torch.randn(b.w_shape, device="cuda") * 0.01).bfloat16()
    x = (torch.randn(b.i_shape(N), device="cuda") * 0.1).bfloat16()
    dy = (torch.randn(b.o_shape(N), device="cuda") * 0.1).bfloat16()
    tf, tb, tu = timeit(lambda: b.fprop(x, w)), timeit(lambda: b.bprop(dy, w)), timeit(lambda: b.updat(x, dy))
    print("cfg3 headline N=%d: fprop %.1f us | bprop %.1f us | updat %.1f us" % (N, tf*1e3, tb*1e3, tu*1e3), flush=True)
