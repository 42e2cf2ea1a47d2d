"""Small-minibatch step (launch-bound): eager launches vs one hipGraph replay (torch.cuda.CUDAGraph captures the ctypes launches)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul
b = BlocksparseMatMul(P.random_layout(128, 128, 0.2, seed=1234), block_size=32, feature_axis=1)
for N in (64, 512):
    w = (torch.randn(b.w_shape, device="cuda") * 0.01).bfloat16()
    x = (torch.randn(b.i_shape(N), device="cuda") * 0.1).bfloat16()
    dy = (torch.randn(b.o_shape(N), device="cuda") * 0.1).bfloat16()
    dw = torch.empty(b.w_shape, dtype=torch.bfloat16, device="cuda")
    def step():
        y = b.fprop(x, w); b.updat(x, dy, dw=dw); dx = b.bprop(dy, w)
        return y, dx
    for _ in range(5): step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        y_g, dx_g = step()
    g.replay(); torch.cuda.synchronize()
    y_e, dx_e = step(); torch.cuda.synchronize()
    ok = torch.equal(y_g, y_e) and torch.equal(dx_g, dx_e)
    def t(fn, reps=200):
        for _ in range(20): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
    print("N=%d: eager %.1f us/step, graph replay %.1f us/step, identical results: %s" % (N, t(step), t(g.replay), ok), flush=True)
