"""Row-owner against windowed weight-gradient kernel (bsize 16, feature axis 0, bf16) over minibatch sizes, densities and splits."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib as lib


def timeit(fn, reps=30):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


_x = torch.randn(8192, 8192, device="cuda"); _t = time.time()
while time.time() - _t < 0.7: _x @ _x
del _x
for nb, dens in ((256, 0.10), (256, 0.05), (256, 0.20), (512, 0.05), (128, 0.10)):
    lay = P.random_layout(nb, nb, dens, seed=1234)
    b_old = BlocksparseMatMul(lay, block_size=16, feature_axis=0, plan_options=lib.PLAN_UPDAT16_WINDOWED)
    objs = {s: BlocksparseMatMul(lay, block_size=16, feature_axis=0, updat_split=s) for s in (1, 2, 4, 8)}
    has = objs[1]._tables_on(torch.device("cuda")).updat_plan.host[8] > 0
    for N in (512, 1024, 2048, 4096, 8192, 16384):
        if nb == 512 and N > 8192: continue
        x = (torch.randn(b_old.i_shape(N), device="cuda") * 0.1).bfloat16()
        e = (torch.randn(b_old.o_shape(N), device="cuda") * 0.1).bfloat16()
        t_old = timeit(lambda: b_old.updat(x, e))
        row = []
        for s, b in objs.items():
            if N // 64 < s: row.append("    -"); continue
            t = timeit(lambda: b.updat(x, e)); k = lib.last_kernel()
            row.append("%6.1f%s" % (t, "" if k == lib.K_UPDAT16_ROWS else "*"))
        print("%d^2 %4.0f%% N=%5d section %s: windowed %6.1f us | rows split 1/2/4/8: %s" % (nb * 16, dens * 100, N, bool(has), t_old, " ".join(row)), flush=True)
