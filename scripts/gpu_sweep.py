"""Measured table for DESIGN.md / profiles: minibatch sweep of the headline workload and the other BASELINE configs.
Prints markdown.  Effective TFLOP/s = 2 * blocks * bs^2 * N / t per pass."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul

def timeit(fn, reps=100):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def row(tag, hidden, bs, dens, axis, dt, N, layout=None):
    td = {"bf16": torch.bfloat16, "f32": torch.float32, "f16": torch.float16}[dt]
    if layout is None:
        layout = P.random_layout(hidden // bs, hidden // bs, dens, seed=1234)
    dens = float(layout.sum()) / layout.size
    b = BlocksparseMatMul(layout, block_size=bs, feature_axis=axis)
    w = (torch.randn(b.w_shape, device="cuda") * 0.01).to(td)
    x = (torch.randn(b.i_shape(N), device="cuda") * 0.1).to(td)
    dy = (torch.randn(b.o_shape(N), device="cuda") * 0.1).to(td)
    fl = 2.0 * b.blocks * bs * bs * N
    s = x.element_size()
    by = s * (2 * hidden * N + b.blocks * bs * bs)
    tf, tb, tu = timeit(lambda: b.fprop(x, w)), timeit(lambda: b.bprop(dy, w)), timeit(lambda: b.updat(x, dy))
    print("| %s | %d | %d | %.0f%% | %d | %s | %d | %d | %.1f / %.0f / %.0f | %.1f / %.0f / %.0f | %.1f / %.0f / %.0f |" % (
        tag, hidden, bs, dens * 100, axis, dt, N, b.blocks, tf * 1e3, fl / tf / 1e9, by / tf / 1e6, tb * 1e3, fl / tb / 1e9, by / tb / 1e6,
        tu * 1e3, fl / tu / 1e9, by / tu / 1e6), flush=True)

# bring the GPU to its boost clock first (a few hundred ms of sustained load)
import time
_b = BlocksparseMatMul(P.random_layout(128, 128, 0.2, seed=1234), block_size=32, feature_axis=1)
_w = torch.zeros(_b.w_shape, device="cuda", dtype=torch.bfloat16); _x = torch.zeros(_b.i_shape(8192), device="cuda", dtype=torch.bfloat16)
_t = time.perf_counter()
while time.perf_counter() - _t < 0.7:
    for _ in range(10): _b.fprop(_x, _w)
    torch.cuda.synchronize()
print("| workload | hidden | bs | density | axis | dtype | N | blocks | fprop us / TF / GB/s | bprop us / TF / GB/s | updat us / TF / GB/s |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for N in (64, 512, 2048, 4096, 8192, 16384):
    row("headline N sweep", 4096, 32, 0.2, 1, "bf16", N)
for d in (0.1, 0.5):
    row("headline density", 4096, 32, d, 1, "bf16", 8192)
row("headline, axis 0", 4096, 32, 0.2, 0, "bf16", 8192)
row("headline, fp16", 4096, 32, 0.2, 1, "f16", 8192)
row("configs[1] fp32", 4096, 32, 0.2, 1, "f32", 8192)
row("configs[1] fp32 axis 0", 4096, 32, 0.2, 0, "f32", 8192)
row("configs[2] bs16", 4096, 16, 0.1, 0, "bf16", 8192)
row("configs[2] bs16 axis 1", 4096, 16, 0.1, 1, "bf16", 8192)
row("configs[3] per-GPU shard", 8192, 32, 0.05, 1, "bf16", 512)
row("bs 8", 4096, 8, 0.1, 0, "bf16", 8192)
row("bs 8 axis 1", 4096, 8, 0.1, 1, "bf16", 8192)
row("bs 8, 3 %", 4096, 8, 0.03, 0, "bf16", 8192)
# skewed layout of about the headline's block count: Barabasi-Albert graph (hubs = block rows / columns with many blocks) + I
# (the reference's own bench layout, test/blocksparse_matmul_bench.py:66-68)
row("headline shape, BA(128, 14) + I layout", 4096, 32, 0.2, 1, "bf16", 8192, layout=P.ba_layout(128, 14, seed=1))
row("same, axis 0", 4096, 32, 0.2, 0, "bf16", 8192, layout=P.ba_layout(128, 14, seed=1))
