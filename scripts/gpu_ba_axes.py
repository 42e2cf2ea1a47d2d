import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib
def timeit(fn, reps=60, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for bs, n, m, dens in ((32, 128, 14, 0.2), (16, 256, 14, 0.1)):
    for name, lay in (("BA", P.ba_layout(n, m, seed=1)), ("uniform", P.random_layout(n, n, P.ba_layout(n, m, seed=1).mean(), 1234))):
        for axis in (0, 1):
            b = BlocksparseMatMul(lay, block_size=bs, feature_axis=axis)
            g = torch.Generator(device="cuda").manual_seed(1)
            w = (torch.randn(b.w_shape, device="cuda", generator=g) * 0.05).bfloat16()
            x = (torch.randn(b.i_shape(8192), device="cuda", generator=g) * 0.1).bfloat16()
            dy = (torch.randn(b.o_shape(8192), device="cuda", generator=g) * 0.1).bfloat16()
            b.fprop(x, w); k = _lib.last_kernel() & 255
            print("bs %d %s (%d blocks) axis %d: k%d fprop %.1f bprop %.1f updat %.1f" % (bs, name, b.blocks, axis, k, timeit(lambda: b.fprop(x, w)), timeit(lambda: b.bprop(dy, w)), timeit(lambda: b.updat(x, dy))), flush=True)
