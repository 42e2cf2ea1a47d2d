import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib as lib
def timeit(fn, reps=40):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
_x = torch.randn(8192, 8192, device="cuda"); _t = time.time()
while time.time() - _t < 0.7: _x @ _x
del _x
print("bs16 axis 0 rows kernel (row stride = 2 N bytes):")
lay = P.random_layout(256, 256, 0.10, seed=1234)
b = BlocksparseMatMul(lay, block_size=16, feature_axis=0, updat_split=4)
for N in (8192, 8256, 8320, 8704):
    x = (torch.randn(b.i_shape(N), device="cuda") * 0.1).bfloat16(); e = (torch.randn(b.o_shape(N), device="cuda") * 0.1).bfloat16()
    t = timeit(lambda: b.updat(x, e)); print("  N=%5d: %.1f us, %.3f us per 64-chunk (kernel %d)" % (N, t, t / (N / 64 / 4), lib.last_kernel()))
print("bs32 axis 1 (row stride = 2 C bytes), 20 %, N = 8192: TF per pass")
for nb in (128, 130, 132):
    lay = P.random_layout(nb, nb, 0.2, seed=1234)
    b = BlocksparseMatMul(lay, block_size=32, feature_axis=1)
    N = 8192
    w = (torch.randn(b.w_shape, device="cuda") * 0.01).bfloat16()
    x = (torch.randn(b.i_shape(N), device="cuda") * 0.1).bfloat16(); e = (torch.randn(b.o_shape(N), device="cuda") * 0.1).bfloat16()
    fl = 2.0 * b.blocks * 1024 * N
    tf, tb, tu = timeit(lambda: b.fprop(x, w)), timeit(lambda: b.bprop(e, w)), timeit(lambda: b.updat(x, e))
    print("  hidden %d (%d blocks): fprop %.1f us = %.0f TF | bprop %.1f us = %.0f TF | updat %.1f us = %.0f TF" % (nb * 32, b.blocks, tf, fl / tf / 1e6, tb, fl / tb / 1e6, tu, fl / tu / 1e6))
