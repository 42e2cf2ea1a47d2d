"""staged plan kernel (forced) vs per-segment kernel (plan ignored) over the minibatch: data for the cost model in xprop_path"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib

def timeit(fn, reps=60, warm=15):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for name, CB, dens in (("4096^2 20%", 128, 0.2), ("4096^2 5%", 128, 0.05), ("8192^2 5%", 256, 0.05), ("2048^2 20%", 64, 0.2)):
    lay = P.random_layout(CB, CB, dens, 1234)
    b = BlocksparseMatMul(lay, block_size=32, feature_axis=1)
    for N in (128, 256, 512, 1024, 2048, 4096, 8192):
        g = torch.Generator(device="cuda").manual_seed(1)
        w = (torch.randn(b.w_shape, device="cuda", generator=g) * 0.05).bfloat16()
        dy = (torch.randn(b.o_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
        out = []
        for v in (3, 2, 0):           # forced plan, no plan, library's choice
            _lib.set_kernel_variant(v)
            b.bprop(dy, w); k = _lib.last_kernel()
            out.append((timeit(lambda: b.bprop(dy, w)), k))
        _lib.set_kernel_variant(0)
        print("%-11s blocks %5d N %5d  plan %.1f (k%d)  segment %.1f (k%d)  auto %.1f (k%d)" % (name, b.blocks, N, out[0][0], out[0][1], out[1][0], out[1][1], out[2][0], out[2][1]), flush=True)
