"""Windowed updat, bsize 32 axis 1: 8x8 vs 16x16 windows (BSMM_UPDAT_WINDOW=8|16 forces one; default = by density)."""
import os, sys, time
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
_x = torch.randn(8192, 8192, device="cuda"); _t = time.time()
while time.time() - _t < 0.7: _x @ _x
tag = os.environ.get("TAG", "")
for hidden, d, Ns in ((4096, 0.05, (8192,)), (4096, 0.10, (2048, 8192)), (4096, 0.15, (8192,)), (4096, 0.20, (8192,)), (8192, 0.05, (512, 4096))):
    b = BlocksparseMatMul(P.random_layout(hidden // 32, hidden // 32, d, seed=1234), block_size=32, feature_axis=1)
    for N in Ns:
        x = (torch.randn(b.i_shape(N), device="cuda") * 0.1).bfloat16()
        dy = (torch.randn(b.o_shape(N), device="cuda") * 0.1).bfloat16()
        tu = timeit(lambda: b.updat(x, dy))
        print("%s updat %d^2 d%.2f N=%d: %.1f us %5.0f TF" % (tag, hidden, d, N, tu * 1e3, 2.0 * b.blocks * 1024 * N / tu / 1e9), flush=True)
