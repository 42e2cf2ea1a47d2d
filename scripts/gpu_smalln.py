"""Small-minibatch regime (per-segment kernels): headline layout at N = 64 / 512 and BASELINE configs[3] (8192^2, 5 %, N = 512)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul
def timeit(fn, reps=200):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for hidden, dens, N, dt in ((4096, 0.2, 64, torch.bfloat16), (4096, 0.2, 512, torch.bfloat16), (8192, 0.05, 512, torch.bfloat16), (4096, 0.2, 64, torch.float32), (4096, 0.2, 512, torch.float32)):
    b = BlocksparseMatMul(P.random_layout(hidden // 32, hidden // 32, dens, seed=1234), block_size=32, feature_axis=1)
    w = (torch.randn(b.w_shape, device="cuda") * 0.01).to(dt)
    x = (torch.randn(b.i_shape(N), device="cuda") * 0.1).to(dt)
    dy = (torch.randn(b.o_shape(N), device="cuda") * 0.1).to(dt)
    fl = 2.0 * b.blocks * 1024 * N
    tf, tb, tu = timeit(lambda: b.fprop(x, w)), timeit(lambda: b.bprop(dy, w)), timeit(lambda: b.updat(x, dy))
    print("%d d%.2f N%-4d %s fprop %.1f us %.0f TF | bprop %.1f us %.0f TF | updat %.1f us %.0f TF" % (hidden, dens, N, str(dt)[6:], tf, fl / tf / 1e6, tb, fl / tb / 1e6, tu, fl / tu / 1e6), flush=True)
