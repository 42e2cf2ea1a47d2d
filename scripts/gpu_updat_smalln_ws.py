import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, time
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib
def graph_us(fn, K=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2): fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(K): fn()
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 / K * 1e6
lay = P.random_layout(256, 256, 0.05, 1234)
for N in (256, 512, 1024, 2048):
    out = []
    for name, opt in (("auto", 0), ("stream16", _lib.PLAN_STREAM_16), ("stream32", _lib.PLAN_STREAM_32)):
        b = BlocksparseMatMul(lay, block_size=32, feature_axis=1, plan_options=opt)
        g = torch.Generator(device="cuda").manual_seed(1)
        x = (torch.randn(b.i_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
        dy = (torch.randn(b.o_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
        dw = torch.empty(b.w_shape, dtype=torch.bfloat16, device="cuda")
        for var in (0, 3):
            _lib.set_kernel_variant(var)
            b.updat(x, dy, dw=dw); k = _lib.last_kernel()
            us = min(graph_us(lambda: b.updat(x, dy, dw=dw)) for _ in range(2))
            out.append("%s/v%d %.1f (k%d)" % (name, var, us, k))
        _lib.set_kernel_variant(0)
    print("N", N, "  ".join(out), flush=True)
