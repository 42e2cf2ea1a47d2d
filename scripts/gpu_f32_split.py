"""fp32 bsize-32 axis-1 xprop: the exact bf16 three-piece kernel (default) vs xcol32f (BSMM_F32_SPLIT=0): time and error vs float64."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul
def timeit(fn, reps=30):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
_x = torch.randn(8192, 8192, device="cuda"); _t = time.time()
while time.time() - _t < 0.7: _x @ _x
tag = os.environ.get("TAG", "")
AX = int(os.environ.get("AX", "1"))
for d in (0.2, 0.1, 0.5):
    lay = P.random_layout(128, 128, d, seed=1234)
    b = BlocksparseMatMul(lay, block_size=32, feature_axis=AX)
    N = 8192
    g = torch.Generator(device="cuda").manual_seed(1)
    w = torch.randn(b.w_shape, device="cuda", generator=g) * 0.01
    x = torch.randn(b.i_shape(N), device="cuda", generator=g) * 0.1
    dy = torch.randn(b.o_shape(N), device="cuda", generator=g) * 0.1
    fl = 2.0 * b.blocks * 1024 * N
    tf, tb = timeit(lambda: b.fprop(x, w)), timeit(lambda: b.bprop(dy, w))
    print(("%s f32 a" + str(AX) + " d%.2f fprop %.3f ms %6.1f TF | bprop %.3f ms %6.1f TF") % (tag, d, tf, fl/tf/1e9, tb, fl/tb/1e9), flush=True)
    if d == 0.2:   # error vs a float64 dense product on a row sample
        rows = slice(0, 256)
        Wd = torch.zeros(4096, 4096, dtype=torch.float64, device="cuda")
        for i, (c, k) in enumerate(b.updat_list):
            Wd[c*32:(c+1)*32, k*32:(k+1)*32] = w[i].double()
        ref = (x[rows].double() @ Wd) if AX == 1 else (Wd.t() @ x[:, rows].double())
        y = (b.fprop(x, w)[rows] if AX == 1 else b.fprop(x, w)[:, rows]).double()
        print("%s    fprop rel l2 error vs float64: %.3e   max |diff| / mean |ref|: %.3e" % (tag, ((y - ref).norm() / ref.norm()).item(), ((y - ref).abs().max() / ref.abs().mean()).item()), flush=True)
