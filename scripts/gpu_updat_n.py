import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib
L = _lib.load()
td = torch.bfloat16
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for dens in (0.2, 0.05):
    b = BlocksparseMatMul(P.random_layout(128, 128, dens, seed=1234), block_size=32, feature_axis=1)
    for N in (64, 256, 512, 1024, 2048, 4096, 8192):
        x = (torch.randn(b.i_shape(N), device="cuda") * 0.1).to(td)
        dy = (torch.randn(b.o_shape(N), device="cuda") * 0.1).to(td)
        line = "d%.2f N%-5d" % (dens, N)
        for v in (0, 2):
            L.bsmm_set_kernel_variant(v)
            t = timeit(lambda: b.updat(x, dy))
            line += " | v%d %.1f us %6.1f TF" % (v, t * 1e3, 2.0 * b.blocks * 1024 * N / t / 1e9)
        L.bsmm_set_kernel_variant(0)
        print(line, flush=True)
