"""per-wave cycle accounting of the flow kernel (needs a build with -DX4_STAMPS: BSMM_LIB=.../libbsmm_stamps.so)"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib as lib
L = lib.load()
lib.set_kernel_variant(3)
d = float(os.environ.get("DENS", "0.2"))
b = BlocksparseMatMul(P.random_layout(128, 128, d, seed=1234), block_size=32, feature_axis=1)
N = 8192
w = (torch.randn(b.w_shape, device="cuda") * 0.01).bfloat16()
dy = (torch.randn(b.o_shape(N), device="cuda") * 0.1).bfloat16()
for _ in range(20): b.bprop(dy, w)
torch.cuda.synchronize()
buf = np.zeros(64 * 16 * 8, dtype=np.uint64)
assert L.bsmm_debug_x4_trace_copy(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(64, 16, 8).astype(np.float64)
names = ["wait fetch/req", "slab poll", "multiply", "progress poll", "request issue", "fetch+progress", "kernel", "events"]
print("density %.2f: cycles per wave over the kernel (mean over 64 WGs x 16 waves | min wave | max wave)" % d)
for k, n in enumerate(names):
    print("  %-16s %9.0f | %9.0f | %9.0f" % (n, t[:, :, k].mean(), t[:, :, k].mean(axis=0).min(), t[:, :, k].mean(axis=0).max()))
acc = t[:, :, :6].sum(axis=2).mean()
print("  accounted %.0f of %.0f (%.0f%%); per wave (WG 0):" % (acc, t[:, :, 6].mean(), 100 * acc / t[:, :, 6].mean()))
for wv in range(16):
    print("   wave %2d: " % wv + " ".join("%7.0f" % t[0, wv, k] for k in range(8)))
