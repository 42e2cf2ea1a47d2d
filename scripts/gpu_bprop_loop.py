"""bprop (or fprop / updat) of the bench shape in a loop, for profiler runs: python scripts/gpu_bprop_loop.py [density %] [bprop|fprop|updat] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul
dens = int(sys.argv[1]) if len(sys.argv) > 1 else 20
side = sys.argv[2] if len(sys.argv) > 2 else "bprop"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
opt = int(os.environ.get("XP_OPT", "0"), 0)
b = BlocksparseMatMul(P.random_layout(128, 128, dens / 100.0, 1234), block_size=32, feature_axis=1, plan_options=opt)
g = torch.Generator(device="cuda").manual_seed(1)
w = (torch.randn(b.w_shape, device="cuda", generator=g) * 0.05).bfloat16()
x = (torch.randn(b.i_shape(8192), device="cuda", generator=g) * 0.1).bfloat16()
dy = (torch.randn(b.o_shape(8192), device="cuda", generator=g) * 0.1).bfloat16()
fn = {"bprop": lambda: b.bprop(dy, w), "fprop": lambda: b.fprop(x, w), "updat": lambda: b.updat(x, dy)}[side]
for _ in range(reps): fn()
torch.cuda.synchronize()
