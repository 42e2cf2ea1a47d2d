"""run the bsize-16 / feature axis 0 weight gradient at BASELINE configs[2] a few times: the workload of the counter passes (gpu_pmc_mem_updat16.sh).
WINDOWED=1: the windowed kernel (plan option PLAN_UPDAT16_WINDOWED) instead of the row-owner kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib as lib
reps = int(os.environ.get("XP_REPS", "6"))
lay = P.random_layout(256, 256, 0.10, seed=1234)
b = BlocksparseMatMul(lay, block_size=16, feature_axis=0, plan_options=lib.PLAN_UPDAT16_WINDOWED if os.environ.get("WINDOWED") else 0)
x = (torch.randn(b.i_shape(8192), device="cuda") * 0.1).bfloat16()
dy = (torch.randn(b.o_shape(8192), device="cuda") * 0.1).bfloat16()
for _ in range(reps):
    b.updat(x, dy)
torch.cuda.synchronize()
print("kernel", lib.last_kernel())
