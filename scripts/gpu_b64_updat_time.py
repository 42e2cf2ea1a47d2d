import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
import _parity as P
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
lay = P.random_layout(64, 64, 0.2, seed=1234)
b = BlocksparseMatMul(lay, block_size=64, feature_axis=1)
N = 8192
x = (torch.randn(b.i_shape(N), device="cuda") * 0.1).bfloat16()
e = (torch.randn(b.o_shape(N), device="cuda") * 0.1).bfloat16()
g = torch.rand(b.blocks, device="cuda")
print("bs64 4096^2 20%% N=8192 updat %.1f us (kernel %d), gated %.1f us" % (timeit(lambda: b.updat(x, e)), lib.last_kernel(), timeit(lambda: b.updat(x, e, gate=g))))
