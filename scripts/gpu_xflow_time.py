"""time the flow kernel (and the staged one for reference) at the bench shape; TAG names the build (ablation builds are wrong by construction)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _parity as P
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
while time.time() - _t < 0.5: _x @ _x
del _x
lib.set_kernel_variant(3)
out = []
for d in [float(v) for v in os.environ.get("DENS", "0.1,0.2,0.5").split(",")]:
    lay = P.random_layout(128, 128, d, seed=1234)
    b4 = BlocksparseMatMul(lay, block_size=32, feature_axis=1)
    N = 8192
    w = (torch.randn(b4.w_shape, device="cuda") * 0.01).bfloat16()
    x = (torch.randn(b4.i_shape(N), device="cuda") * 0.1).bfloat16()
    dy = (torch.randn(b4.o_shape(N), device="cuda") * 0.1).bfloat16()
    s = "d%.2f flow f %.1f b %.1f" % (d, timeit(lambda: b4.fprop(x, w)), timeit(lambda: b4.bprop(dy, w)))
    assert lib.last_kernel() == lib.K_XCOL32_FLOW
    if os.environ.get("REF"):
        b2 = BlocksparseMatMul(lay, block_size=32, feature_axis=1); b2.flow = False
        s += " | staged f %.1f b %.1f" % (timeit(lambda: b2.fprop(x, w)), timeit(lambda: b2.bprop(dy, w)))
    out.append(s)
print("%-14s %s" % (os.environ.get("TAG", os.path.basename(os.environ.get("BSMM_LIB", "default"))), "  ||  ".join(out)), flush=True)
