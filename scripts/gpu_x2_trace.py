"""Cycle-level breakdown of xcol32_a1_v2_kernel rows (needs a build with -DBSMM_XC_TRACE; BSMM_LIB selects it)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib
d = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
b = BlocksparseMatMul(P.random_layout(128, 128, d, seed=1234), block_size=32, feature_axis=1)
N = 8192
w = (torch.randn(b.w_shape, device="cuda") * 0.01).bfloat16()
dy = (torch.randn(b.o_shape(N), device="cuda") * 0.1).bfloat16()
for _ in range(3): b.bprop(dy, w)
torch.cuda.synchronize()
assert _lib.last_kernel() == _lib.K_XCOL32_STAGED
L = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(8 * 16 * 48 * 5, dtype=np.uint64)
assert L.bsmm_debug_trace_copy2(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(8, 16, 48, 5).astype(np.int64)
ph = t[:, :, 8:44, :]
names = ["wait vmcnt", "barrier", "requests", "blocks"]
tot = ph[:, :, 1:, 0] - ph[:, :, :-1, 0]
print("row period: mean %.0f clk (min %d max %d)" % (tot.mean(), tot.min(), tot.max()))
for k, nm in enumerate(names):
    x = ph[..., k + 1] - ph[..., k]
    print("  %-12s mean %7.0f  p10 %6.0f  p50 %6.0f  p90 %6.0f  max %6.0f" % (nm, x.mean(), np.percentile(x, 10), np.percentile(x, 50), np.percentile(x, 90), x.max()))
gap = ph[:, :, 1:, 0] - ph[:, :, :-1, 4]
print("  %-12s mean %7.0f" % ("loop gap", gap.mean()))
print("workgroup 0, rows 20..23, per wave [wait, barrier, requests, blocks]:")
for row in range(20, 24):
    for v in range(16):
        x = t[0, v, row]
        print("   row %d wave %2d: %5d %5d %5d %5d" % (row, v, x[1] - x[0], x[2] - x[1], x[3] - x[2], x[4] - x[3]))
