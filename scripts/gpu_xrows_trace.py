"""per-wave cycle accounting of the row-split kernel (needs a build with -DX5_STAMPS: BSMM_LIB=.../libbsmm_x5stamps.so)"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib as lib
L = lib.load()
lib.set_kernel_variant(3)
d = float(os.environ.get("DENS", "0.2"))
b = BlocksparseMatMul(P.random_layout(128, 128, d, seed=1234), block_size=32, feature_axis=1); b.rows = True
N = 8192
w = (torch.randn(b.w_shape, device="cuda") * 0.01).bfloat16()
dy = (torch.randn(b.o_shape(N), device="cuda") * 0.1).bfloat16()
for _ in range(20): b.bprop(dy, w)
torch.cuda.synchronize()
assert lib.last_kernel() == lib.K_XCOL32_ROWS
buf = np.zeros(64 * 4 * 8, dtype=np.uint64)
assert L.bsmm_debug_x5_trace_copy(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(64, 4, 8).astype(np.float64)
names = ["record -> count", "vmcnt wait", "barrier", "first + early duties", "blocks", "record + late duties", "unit end", "kernel"]
steps = 2 * 62.0
print("%s density %.2f: cycles per wave over the kernel (mean over 64 WGs x 4 waves | per step of ~124)" % (os.environ.get("TAG", ""), d))
for k, n in enumerate(names):
    print("  %-22s %9.0f | %7.0f" % (n, t[:, :, k].mean(), t[:, :, k].mean() / steps))
