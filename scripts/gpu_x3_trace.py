"""Cycle-level breakdown of the phases of xcol32_v3_kernel (needs a build with -DBSMM_X3_TRACE): python scripts/gpu_x3_trace.py [density %] [fprop|bprop]"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib
dens = int(sys.argv[1]) if len(sys.argv) > 1 else 20
side = sys.argv[2] if len(sys.argv) > 2 else "bprop"
opt = int(os.environ.get("XP_OPT", "0"), 0)
b = BlocksparseMatMul(P.random_layout(128, 128, dens / 100.0, seed=1234), block_size=32, feature_axis=1, plan_options=opt)
N = 8192
w = (torch.randn(b.w_shape, device="cuda") * 0.01).bfloat16()
dy = (torch.randn(b.o_shape(N), device="cuda") * 0.1).bfloat16()
x = (torch.randn(b.i_shape(N), device="cuda") * 0.1).bfloat16()
fn = (lambda: b.bprop(dy, w)) if side == "bprop" else (lambda: b.fprop(x, w))
for _ in range(30): fn()
torch.cuda.synchronize()
L = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(8 * 16 * 48 * 6, dtype=np.uint64)
assert L.bsmm_debug_x3_trace_copy(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(8, 16, 48, 6).astype(np.int64)
ph = t[:, :, 4:28, :]                      # steady-state phases
names = ["wait vmcnt (my shares landed)", "barrier", "weight requests", "blocks (reads + MFMA)", "slab requests"]
tot = ph[:, :, 1:, 0] - ph[:, :, :-1, 0]
print("%s %d %% opt %#x: phase period mean %.0f clk (min %d max %d)" % (side, dens, opt, tot.mean(), tot.min(), tot.max()))
for k, nm in enumerate(names):
    x_ = ph[..., k + 1] - ph[..., k]
    print("  %-34s mean %7.0f  p10 %6.0f  p50 %6.0f  p90 %6.0f  max %6.0f" % (nm, x_.mean(), np.percentile(x_, 10), np.percentile(x_, 50), np.percentile(x_, 90), x_.max()))
gap = ph[:, :, 1:, 0] - ph[:, :, :-1, 5]
print("  %-34s mean %7.0f" % ("loop overhead (end -> next top)", gap.mean()))
# who arrives last at the barrier: time from the phase's first barrier release to each wave's arrival at the NEXT wait
print("one workgroup, phases 10..13, per wave [wait, barrier, wreq, blocks, xreq]:")
for q in range(10, 14):
    for v in range(16):
        x_ = t[0, v, q]
        print("   phase %d wave %2d: %5d %5d %5d %5d %5d" % (q, v, x_[1] - x_[0], x_[2] - x_[1], x_[3] - x_[2], x_[4] - x_[3], x_[5] - x_[4]))
