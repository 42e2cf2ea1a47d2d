"""fp32 xprop on feature axis 1 (BASELINE configs[1]): the kernel with the activation split fused in (default build) against the
pre-pass form (a build with -DXS_NO_FUSE=1, BSMM_LIB=...): prints a digest of every output (compare the two runs: the forms are
bit-identical), the worst L2 error against the float64 oracle on sampled columns, and timings."""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib as lib

def timeit(fn, reps=50):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

def digest(t):
    return hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:16]

print("library", os.path.basename(os.environ.get("BSMM_LIB", "default")), flush=True)
lib.set_kernel_variant(3)
cases = [("128x128 20% N=1024", P.random_layout(128, 128, 0.2, seed=1234), 1024),
         ("33x40 30% N=520 (odd input blocks, ragged rows)", P.random_layout(33, 40, 0.3, seed=3), 520),
         ("40x33 30% N=136 (odd output blocks)", P.random_layout(40, 33, 0.3, seed=4), 136),
         ("BA 64 N=256", P.ba_layout(64, 5, seed=1), 256),
         ("200x16 10% N=384 (> 64 steps in a group)", P.random_layout(200, 16, 0.1, seed=6), 384),
         ("1x1 N=40", np.ones((1, 1), dtype=np.int32), 40)]
for name, lay, N in cases:
    b = BlocksparseMatMul(lay, block_size=32, feature_axis=1)
    g = torch.Generator(device="cuda").manual_seed(11)
    w = torch.randn(b.w_shape, device="cuda", generator=g) * 0.05
    x = torch.randn(b.i_shape(N), device="cuda", generator=g)
    dy = torch.randn(b.o_shape(N), device="cuda", generator=g)
    y = b.fprop(x, w); kf = lib.last_kernel()
    dx = b.bprop(dy, w); kb = lib.last_kernel()
    torch.cuda.synchronize()
    from oracle import bsmm_oracle as orc
    t = orc.build_layout_luts(np.asarray(lay), 32)
    yr = orc.fprop(t, x.cpu().numpy(), w.cpu().numpy(), 1); dxr = orc.bprop(t, dy.cpu().numpy(), w.cpu().numpy(), 1)
    l2y = np.linalg.norm(y.double().cpu().numpy() - yr) / max(np.linalg.norm(yr), 1e-30)
    l2x = np.linalg.norm(dx.double().cpu().numpy() - dxr) / max(np.linalg.norm(dxr), 1e-30)
    print("case %-48s kernels %d/%d  y %s dx %s  L2 vs float64 %.2e / %.2e" % (name, kf, kb, digest(y), digest(dx), l2y, l2x), flush=True)
lib.set_kernel_variant(0)
_x = torch.randn(8192, 8192, device="cuda"); _t = time.time()
while time.time() - _t < 0.7: _x @ _x
del _x
for d in (0.1, 0.2, 0.5):
    lay = P.random_layout(128, 128, d, seed=1234)
    b = BlocksparseMatMul(lay, block_size=32, feature_axis=1)
    N = 8192
    w = torch.randn(b.w_shape, device="cuda") * 0.05
    x = torch.randn(b.i_shape(N), device="cuda"); dy = torch.randn(b.o_shape(N), device="cuda")
    def f():
        w.add_(0)                      # a new weights version per call: what a training step pays
        return b.fprop(x, w)
    def g_():
        w.add_(0)
        return b.bprop(dy, w)
    tf, tb = timeit(f), timeit(g_)
    tfc, tbc = timeit(lambda: b.fprop(x, w)), timeit(lambda: b.bprop(dy, w))
    fl = 2.0 * b.blocks * 1024 * N
    print("d%.2f fprop %.1f us (%.0f TF; weights cached %.1f)  bprop %.1f us (%.0f TF; cached %.1f)  y %s" %
          (d, tf, fl / tf * 1e-6, tfc, tb, fl / tb * 1e-6, tbc, digest(b.fprop(x, w))), flush=True)
