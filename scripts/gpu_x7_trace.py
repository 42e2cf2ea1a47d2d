"""Cycle-level breakdown of the phases of xcol16_list_kernel at BASELINE configs[2] (needs a build with -DX7L_TRACE, see build_variants.py)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib
axis = int(sys.argv[1]) if len(sys.argv) > 1 else 0
b = BlocksparseMatMul(P.random_layout(256, 256, 0.1, seed=1234), block_size=16, feature_axis=axis)
N = 8192
w = (torch.randn(b.w_shape, device="cuda") * 0.01).bfloat16()
dy = (torch.randn(b.o_shape(N), device="cuda") * 0.1).bfloat16()
for _ in range(3): b.bprop(dy, w)
torch.cuda.synchronize()
L = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(8 * 16 * 40 * 5, dtype=np.uint64)
assert L.bsmm_debug_x7_trace_copy(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(8, 16, 40, 5).astype(np.int64)
ph = t[:, :, 2:30, :]
names = ["wait vmcnt(0)", "barrier", "first block + requests", "rest of the list"]
tot = ph[:, :, 1:, 0] - ph[:, :, :-1, 0]
print("axis %d bprop: phase period mean %.0f clk (min %d max %d)" % (axis, tot.mean(), tot.min(), tot.max()))
for k, nm in enumerate(names):
    x = ph[..., k + 1] - ph[..., k]
    print("  %-26s mean %6.0f  p10 %6.0f  p50 %6.0f  p90 %6.0f  max %6.0f" % (nm, x.mean(), np.percentile(x, 10), np.percentile(x, 50), np.percentile(x, 90), x.max()))
print("  %-26s mean %6.0f" % ("loop overhead", (ph[:, :, 1:, 0] - ph[:, :, :-1, 4]).mean()))
busy = (ph[..., 4] - ph[..., 2])
print("  per phase: busiest wave (barrier -> end of list) mean %.0f, mean wave %.0f" % (busy.max(axis=1).mean(), busy.mean()))
print("one workgroup, phases 10..11, per wave [wait, barrier, first+req, rest]:")
for p in (10, 11):
    for v in range(16):
        x = t[0, v, p]
        print("   phase %d wave %2d: %5d %5d %5d %5d" % (p, v, x[1] - x[0], x[2] - x[1], x[3] - x[2], x[4] - x[3]))
