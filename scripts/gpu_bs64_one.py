import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib as lib
def timeit(fn, reps=50):
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
for bs, nb in ((64, 64), (32, 128)):
    lay = P.random_layout(nb, nb, 0.2, seed=1234)
    b = BlocksparseMatMul(lay, block_size=bs, feature_axis=1)
    N = 8192
    w = (torch.randn(b.w_shape, device="cuda") * 0.01).bfloat16()
    x = (torch.randn(b.i_shape(N), device="cuda") * 0.1).bfloat16()
    dy = (torch.randn(b.o_shape(N), device="cuda") * 0.1).bfloat16()
    tf = timeit(lambda: b.fprop(x, w)); kf = lib.last_kernel()
    tb = timeit(lambda: b.bprop(dy, w)); kb = lib.last_kernel()
    tu = timeit(lambda: b.updat(x, dy)); ku = lib.last_kernel()
    fl = 2.0 * b.blocks * bs * bs * N
    print("bs%d blocks %d: fprop %.1f us (%.0f TF, k%d) bprop %.1f (%.0f TF, k%d) updat %.1f (%.0f TF, k%d)" % (bs, b.blocks, tf, fl/tf*1e-6, kf, tb, fl/tb*1e-6, kb, tu, fl/tu*1e-6, ku), flush=True)
