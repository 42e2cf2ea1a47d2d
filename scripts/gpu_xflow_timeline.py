"""time line of the flow kernel's workgroup 0 (needs -DX4_TIMELINE): per step, when the ring slot became free, when its parts were
requested / announced, when the blocks got the slab and finished"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib as lib
L = lib.load()
lib.set_kernel_variant(3)
d = float(os.environ.get("DENS", "0.2")); D = int(os.environ.get("D", "5"))
b = BlocksparseMatMul(P.random_layout(128, 128, d, seed=1234), block_size=32, feature_axis=1)
N = 8192
w = (torch.randn(b.w_shape, device="cuda") * 0.01).bfloat16()
dy = (torch.randn(b.o_shape(N), device="cuda") * 0.1).bfloat16()
for _ in range(5): b.bprop(dy, w)
torch.cuda.synchronize()
buf = np.zeros(256 * 40, dtype=np.uint64)
assert L.bsmm_debug_x4_timeline_copy(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(256, 40).astype(np.int64)
t0 = t[t > 0].min()
t = np.where(t > 0, t - t0, -1)
print("step | slot free (last block of step-D done) | req start p0 p1 | req issued p0 p1 | ann p0 p1 | first block ready | last block done | #blocks")
for s in range(int(os.environ.get("S0", "16")), int(os.environ.get("S1", "64"))):
    done = t[s, 7:39:2]; ready = t[s, 6:38:2]
    prev = t[s - D, 7:39:2] if s >= D else np.array([-1])
    free = prev.max() if (prev >= 0).any() else -1
    print("%4d | %7d | %7d %7d | %7d %7d | %7d %7d | %7d | %7d | %d" % (s, free, t[s, 0], t[s, 1], t[s, 2], t[s, 3], t[s, 4], t[s, 5],
          ready[ready >= 0].min() if (ready >= 0).any() else -1, done.max(), int((done >= 0).sum())))
if os.environ.get("DETAIL"):
    a, bb = [int(v) for v in os.environ["DETAIL"].split(",")]
    print("per wave: (ready, done) of its block in each step; blank = no block")
    print("wave " + " ".join("      step %3d     " % s for s in range(a, bb)))
    for wv in range(16):
        print("%4d " % wv + " ".join(("%8d %8d " % (t[s, 6 + 2 * wv], t[s, 7 + 2 * wv])) if t[s, 7 + 2 * wv] >= 0 else " " * 18 for s in range(a, bb)))
