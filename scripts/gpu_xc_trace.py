"""Cycle-level breakdown of xcol32_a1_kernel phases (needs a build with -DBSMM_XC_TRACE)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib
b = BlocksparseMatMul(P.random_layout(128, 128, 0.2, seed=1234), block_size=32, feature_axis=1)
N = 8192
w = (torch.randn(b.w_shape, device="cuda") * 0.01).bfloat16()
dy = (torch.randn(b.o_shape(N), device="cuda") * 0.1).bfloat16()
for _ in range(3): b.bprop(dy, w)
torch.cuda.synchronize()
L = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(8 * 8 * 40 * 6, dtype=np.uint64)
assert L.bsmm_debug_trace_copy(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(8, 8, 40, 6).astype(np.int64)
ph = t[:, :, 2:30, :]                      # steady-state phases
names = ["wait vmcnt(0) (slab+W landed)", "barrier", "DMA issue (4 instr)", "step 0 (W loads + blocks)", "wait W of step 1", "step 1 (blocks)"]
d = [ph[..., 1] - ph[..., 0], ph[..., 2] - ph[..., 1], ph[..., 3] - ph[..., 2], ph[..., 4] - ph[..., 3], None, ph[..., 5] - ph[..., 4]]
tot = ph[:, :, 1:, 0] - ph[:, :, :-1, 0]
print("phase period: mean %.0f clk (min %d max %d)" % (tot.mean(), tot.min(), tot.max()))
for nm, x in zip(names, d):
    if x is not None:
        print("  %-34s mean %7.0f  p10 %6.0f  p90 %6.0f" % (nm, x.mean(), np.percentile(x, 10), np.percentile(x, 90)))
gap = ph[:, :, 1:, 0] - ph[:, :, :-1, 5]
print("  %-34s mean %7.0f" % ("loop overhead (end -> next top)", gap.mean()))
# per-wave view of one workgroup / phase
print("one workgroup, phase 10, per wave [wait, barrier, dma, step0, step1]:")
for v in range(8):
    x = t[0, v, 10]
    print("   wave %d: %5d %5d %5d %5d %5d" % (v, x[1] - x[0], x[2] - x[1], x[3] - x[2], x[4] - x[3], x[5] - x[4]))
