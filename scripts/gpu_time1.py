"""Time one pass for a list of BSMM_DBG values (ablation).  env as gpu_one.py + DBGS=0,1,2,..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib
e = os.environ.get
what, axis, bs, dens, N = e("PASS", "bprop"), int(e("AXIS", "1")), int(e("BS", "32")), float(e("DENS", "0.2")), int(e("N", "8192"))
td = torch.bfloat16
H = int(e("HIDDEN", "4096"))
b = BlocksparseMatMul(P.random_layout(H // bs, H // bs, dens, seed=1234), block_size=bs, feature_axis=axis)
w = (torch.randn(b.w_shape, device="cuda") * 0.01).to(td)
x = (torch.randn(b.i_shape(N), device="cuda") * 0.1).to(td)
dy = (torch.randn(b.o_shape(N), device="cuda") * 0.1).to(td)
fn = {"fprop": lambda: b.fprop(x, w), "bprop": lambda: b.bprop(dy, w), "updat": lambda: b.updat(x, dy)}[what]
for d in e("DBGS", "0").split(","):
    os.environ["BSMM_DBG"] = d
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    print("dbg=%-3s %s %.1f us  %.1f TF" % (d, what, t * 1e3, 2.0 * b.blocks * bs * bs * N / t / 1e9), flush=True)
