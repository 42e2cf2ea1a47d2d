"""Time the bsize-8 passes (4096^2, 10 %, N=8192, bf16): super-block path (default) vs the V_FMA kernels (variant 1)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blocksparse_amd import BlocksparseMatMul, _lib

def layout(n, d, seed=1234):
    rng = np.random.default_rng(seed)
    return (rng.random((n, n)) < d).astype(np.int32)

def t(fn, it=30, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it

L = _lib.load()
x0 = torch.randn(8192, 8192, device="cuda"); t0 = time.time()
while time.time() - t0 < 0.7: x0 @ x0
for dens in (0.10, 0.03):
    for axis in (0, 1):
        N, H = 8192, 4096
        b = BlocksparseMatMul(layout(H // 8, dens), block_size=8, feature_axis=axis)
        w = (torch.randn(b.w_shape, device="cuda") * 0.02).bfloat16()
        x = torch.randn(b.i_shape(N), device="cuda").bfloat16()
        e = torch.randn(b.o_shape(N), device="cuda").bfloat16()
        fl = 2.0 * b.blocks * 64 * N
        for variant, name in ((0, "super"), (1, "valu")):
            L.bsmm_set_kernel_variant(variant)
            tf, tb, tu = t(lambda: b.fprop(x, w)), t(lambda: b.bprop(e, w)), t(lambda: b.updat(x, e))
            print("bs8 d%.2f a%d %-5s fprop %.3f ms %6.1f TF | bprop %.3f ms %6.1f TF | updat %.3f ms %6.1f TF" % (
                dens, axis, name, tf, fl / tf / 1e9, tb, fl / tb / 1e9, tu, fl / tu / 1e9), flush=True)
        L.bsmm_set_kernel_variant(0)
