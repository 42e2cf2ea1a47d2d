import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib as lib
def timeit(fn, reps=100):
    for _ in range(15): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
_x = torch.randn(8192, 8192, device="cuda"); _t = time.time()
while time.time() - _t < 0.7: _x @ _x
del _x
for d, N in ((0.1, 8192), (0.2, 8192), (0.05, 8192), (0.1, 2048), (0.2, 2048)):
    lay = P.random_layout(128, 128, d, seed=1234)
    x = (torch.randn(N, 4096, device="cuda") * 0.1).bfloat16(); dy = (torch.randn(N, 4096, device="cuda") * 0.1).bfloat16()
    row = []
    for sets in (0, 1, 2, 4, 8):
        b = BlocksparseMatMul(lay, block_size=32, feature_axis=1, plan_options=sets << 12)
        lib.set_kernel_variant(3)
        t1 = timeit(lambda: b.updat(x, dy)); t2 = timeit(lambda: b.updat(x, dy))
        lib.set_kernel_variant(0)
        row.append("sets %d: %.1f/%.1f" % (sets, t1, t2))
    print("d%.2f N%d  " % (d, N) + "  ".join(row), flush=True)
