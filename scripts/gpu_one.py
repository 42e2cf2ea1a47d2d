"""Run ONE pass/config repeatedly (for rocprofv3 --pmc / --kernel-trace).  env: PASS=fprop|bprop|updat AXIS BS DENS N DTYPE VARIANT REPS"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib

e = os.environ.get
what, axis, bs, dens, N = e("PASS", "bprop"), int(e("AXIS", "1")), int(e("BS", "32")), float(e("DENS", "0.2")), int(e("N", "8192"))
td = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[e("DTYPE", "bf16")]
_lib.load().bsmm_set_kernel_variant(int(e("VARIANT", "0")))
CB = 4096 // bs
b = BlocksparseMatMul(P.random_layout(CB, CB, dens, seed=1234), block_size=bs, feature_axis=axis)
w = (torch.randn(b.w_shape, device="cuda") * 0.01).to(td)
x = (torch.randn(b.i_shape(N), device="cuda") * 0.1).to(td)
dy = (torch.randn(b.o_shape(N), device="cuda") * 0.1).to(td)
fn = {"fprop": lambda: b.fprop(x, w), "bprop": lambda: b.bprop(dy, w), "updat": lambda: b.updat(x, dy)}[what]
for _ in range(int(e("REPS", "5"))):
    fn()
torch.cuda.synchronize()
print("done", what, axis, bs, dens, N)
