"""per-wave cycle accounting of the row-owner weight-gradient kernel (needs a build with -DU6_STAMPS: BSMM_LIB=.../libbsmm_u6stamps.so)"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib as lib
L = lib.load()
split = int(os.environ.get("SPLIT", "4"))
b = BlocksparseMatMul(P.random_layout(256, 256, 0.10, seed=1234), block_size=16, feature_axis=0, updat_split=split)
N = 8192
x = (torch.randn(b.i_shape(N), device="cuda") * 0.1).bfloat16()
e = (torch.randn(b.o_shape(N), device="cuda") * 0.1).bfloat16()
for _ in range(20): b.updat(x, e)
torch.cuda.synchronize()
assert lib.last_kernel() == lib.K_UPDAT16_ROWS
buf = np.zeros(64 * 16 * 8, dtype=np.uint64)
assert L.bsmm_debug_u6_trace_copy(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(64, 16, 8).astype(np.float64)
names = ["request next chunk", "counted wait", "barrier (data)", "matrix work", "barrier (slot free)", "epilogue", "kernel", "chunks"]
ch = t[:, :, 7].mean()
print("%s split %d: cycles per wave (mean over 64 WGs x 16 waves | per chunk of %.0f | min .. max of the per-wave means over the workgroups)" % (os.environ.get("TAG", ""), split, ch))
for k, n in enumerate(names):
    pw = t[:, :, k].mean(axis=0)
    print("  %-22s %9.0f | %7.0f | %7.0f .. %7.0f" % (n, t[:, :, k].mean(), t[:, :, k].mean() / ch, pw.min() / ch, pw.max() / ch))
