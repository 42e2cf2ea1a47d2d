"""Cost of the minibatch-split (fp32 atomics + finalize) path of the windowed updat at the headline size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul
def timeit(fn, reps=100):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
b = BlocksparseMatMul(P.random_layout(128, 128, 0.2, seed=1234), block_size=32, feature_axis=1)
N = 8192
x = (torch.randn(b.i_shape(N), device="cuda") * 0.1).bfloat16()
dy = (torch.randn(b.o_shape(N), device="cuda") * 0.1).bfloat16()
t = timeit(lambda: b.updat(x, dy))
print("BSMM_UPDAT_SPLIT=%s updat %.1f us" % (os.environ.get("BSMM_UPDAT_SPLIT", "auto"), t * 1e3))
