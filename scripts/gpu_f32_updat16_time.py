import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib as lib
def timeit(fn, reps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
_x = torch.randn(8192, 8192, device="cuda"); _t = time.time()
while time.time() - _t < 0.7: _x @ _x
del _x
lay = P.random_layout(256, 256, 0.10, seed=1234)
N = 8192
for opt, name in ((0, "row-owner path"), (lib.PLAN_UPDAT16_WINDOWED, "per-block fp32 kernel")):
    b = BlocksparseMatMul(lay, block_size=16, feature_axis=0, plan_options=opt)
    x = torch.randn(b.i_shape(N), device="cuda") * 0.1
    e = torch.randn(b.o_shape(N), device="cuda") * 0.1
    t = timeit(lambda: b.updat(x, e)); k = lib.last_kernel()
    print("fp32 bs16 axis 0 4096^2 10%% N=8192 updat, %s: %.1f us = %.1f TF (kernel %d)" % (name, t, 2.0 * b.blocks * 256 * N / t / 1e6, k))
