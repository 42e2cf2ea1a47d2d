"""Effective TFLOP/s of the three passes over the (block size, feature axis, dtype) matrix at 4096 x 4096, N = 8192 (20 % density for
bsize 32 / 64, 10 % for 16 / 8): which kernel family ran and how fast -- to find the paths that lag."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib as lib

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

_x = torch.randn(8192, 8192, device="cuda"); _t = time.time()
while time.time() - _t < 0.7: _x @ _x
del _x
N = int(os.environ.get("N", "8192"))
for bs in (32, 64, 16, 8):
    dens = 0.2 if bs >= 32 else 0.1
    nb = 4096 // bs
    lay = P.random_layout(nb, nb, dens, seed=1234)
    for axis in (1, 0):
        if bs == 64 and axis == 0:
            continue
        for td in (torch.bfloat16, torch.float32):
            b = BlocksparseMatMul(lay, block_size=bs, feature_axis=axis)
            w = (torch.randn(b.w_shape, device="cuda") * 0.01).to(td)
            x = (torch.randn(b.i_shape(N), device="cuda") * 0.1).to(td)
            dy = (torch.randn(b.o_shape(N), device="cuda") * 0.1).to(td)
            fl = 2.0 * b.blocks * bs * bs * N
            out = []
            for name, fn in (("fprop", lambda: b.fprop(x, w)), ("bprop", lambda: b.bprop(dy, w)), ("updat", lambda: b.updat(x, dy))):
                try:
                    t = timeit(fn)
                    out.append("%s %7.1f us %6.1f TF k%-3d" % (name, t, fl / t * 1e-6, lib.last_kernel()))
                except Exception as ex:
                    out.append("%s FAILED %s" % (name, str(ex)[:60]))
            print("bs%-2d axis %d %-8s d%.0f%%: %s" % (bs, axis, str(td).split(".")[-1], dens * 100, " | ".join(out)), flush=True)
            del b, w, x, dy
