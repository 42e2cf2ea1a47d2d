"""minibatch sweep of the three passes (production dispatch) at the headline layout and BASELINE configs[3]'s layout, timed as hipGraph
replays of 20 back-to-back calls (the eager Python call path costs ~10-15 us per call: at small minibatches it would be all one sees);
TAG names the build"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib as lib
K = 20
def graph_time(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(K): fn()
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 / K * 1e6
def eager_time(fn, reps=200):
    for _ in range(30): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
tag = os.environ.get("TAG", os.path.basename(os.environ.get("BSMM_LIB", "default")))
for hidden, dens in ((4096, 0.2), (8192, 0.05)):
    b = BlocksparseMatMul(P.random_layout(hidden // 32, hidden // 32, dens, seed=1234), block_size=32, feature_axis=1)
    w = (torch.randn(b.w_shape, device="cuda") * 0.01).bfloat16()
    for N in [int(v) for v in os.environ.get("NS", "64,128,256,512,1024,2048").split(",")]:
        x = (torch.randn(b.i_shape(N), device="cuda") * 0.1).bfloat16()
        dy = (torch.randn(b.o_shape(N), device="cuda") * 0.1).bfloat16()
        dw = torch.empty(b.w_shape, dtype=torch.bfloat16, device="cuda")
        tf = graph_time(lambda: b.fprop(x, w)); kf = lib.last_kernel()
        tb = graph_time(lambda: b.bprop(dy, w)); kb = lib.last_kernel()
        tu = graph_time(lambda: b.updat(x, dy, dw=dw)); ku = lib.last_kernel() + 100 * lib.last_kernel_variant()
        te = eager_time(lambda: b.bprop(dy, w))
        print("%-16s %d d%.2f N%-5d fprop %6.1f us (k%d) | bprop %6.1f us (k%d) | updat %6.1f us (k%d) | eager bprop %5.1f" % (tag, hidden, dens, N, tf, kf, tb, kb, tu, ku, te), flush=True)
