"""run the row-split kernel (fprop + bprop, bsize 32, axis 1, bf16, N = 8192) a few times: the workload of the counter passes (gpu_pmc_xrows.sh)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib as lib
d = float(sys.argv[1]) / 100.0 if len(sys.argv) > 1 else 0.2
reps = int(os.environ.get("XP_REPS", "10"))
lay = P.random_layout(128, 128, d, seed=1234)
b = BlocksparseMatMul(lay, block_size=32, feature_axis=1); b.rows = True
if os.environ.get("FLOW"): b.rows = False
w = (torch.randn(b.w_shape, device="cuda") * 0.01).bfloat16()
x = (torch.randn(b.i_shape(8192), device="cuda") * 0.1).bfloat16()
dy = (torch.randn(b.o_shape(8192), device="cuda") * 0.1).bfloat16()
for _ in range(reps):
    b.fprop(x, w); b.bprop(dy, w)
torch.cuda.synchronize()
print("kernel", lib.last_kernel())
