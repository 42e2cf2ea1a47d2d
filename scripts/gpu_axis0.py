import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib
td = torch.bfloat16
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
lay = P.ba_layout(40, 3, seed=1)
for N in (64, 104, 328):
    r = P.run_case(torch, BlocksparseMatMul, lay, 32, 0, "bf16", N, seed=3)
    print("parity a0 N%d" % N, " ".join("%s=%.1e" % (k, v[0]) for k, v in r.items()), "FAIL" if any(v[0] > 1e-3 for v in r.values()) else "ok")
for dens in (0.2, 0.5):
    b = BlocksparseMatMul(P.random_layout(128, 128, dens, seed=1234), block_size=32, feature_axis=0)
    for N in (512, 8192):
        w = (torch.randn(b.w_shape, device="cuda") * 0.01).to(td)
        x = (torch.randn(b.i_shape(N), device="cuda") * 0.1).to(td)
        dy = (torch.randn(b.o_shape(N), device="cuda") * 0.1).to(td)
        fl = 2.0 * b.blocks * 1024 * N
        tf, tb, tu = timeit(lambda: b.fprop(x, w)), timeit(lambda: b.bprop(dy, w)), timeit(lambda: b.updat(x, dy))
        print("a0 d%.2f N%-5d fprop %.3f ms %6.1f TF | bprop %.3f ms %6.1f TF | updat %.3f ms %6.1f TF" % (dens, N, tf, fl/tf/1e9, tb, fl/tb/1e9, tu, fl/tu/1e9), flush=True)
