"""Flow kernel ('BSX4', bsmm_xflow.h) against the staged kernel ('BSX2'): bit identity on a set of shapes, then timing at the bench
workload.  Run under `timeout` (a synchronisation bug in a barrier-free kernel shows as a hang): every case prints before it runs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib as lib

def timeit(fn, reps=100):
    for _ in range(15): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

def pair(layout):
    b2 = BlocksparseMatMul(layout, block_size=32, feature_axis=1)
    b2.flow = False
    return (b2, BlocksparseMatMul(layout, block_size=32, feature_axis=1))

def balanced(layout):
    return BlocksparseMatMul(layout, block_size=32, feature_axis=1, plan_options=lib.PLAN_FLOW_SCHEDULED)

cases = [("tiny 4x4 N=128", P.random_layout(4, 4, 0.6, seed=1), 128, torch.bfloat16),
         ("40x24 N=1000 (ragged rows, partial group)", P.random_layout(40, 24, 0.3, seed=2), 1000, torch.bfloat16),
         ("33x35 N=520 (odd block counts)", P.random_layout(33, 35, 0.25, seed=3), 520, torch.float16),
         ("128x128 20% N=2048", P.random_layout(128, 128, 0.2, seed=1234), 2048, torch.bfloat16),
         ("128x128 50% N=1024 (lists > 64 blocks)", P.random_layout(128, 128, 0.55, seed=5), 1024, torch.bfloat16),
         ("300x16 5% N=640 (> 64 steps)", P.random_layout(300, 16, 0.05, seed=6), 640, torch.bfloat16),
         ("BA 128 N=4096", P.ba_layout(128, 14, seed=1), 4096, torch.bfloat16),
         ("15x33 with an output group without blocks, N=520", np.eye(15, 33, dtype=np.int32), 520, torch.float16)]
lib.set_kernel_variant(3)
ok = True
for name, lay, N, td in cases:
    print("case", name, flush=True)
    b2, b4 = pair(lay)
    g = torch.Generator(device="cuda").manual_seed(7)
    w = (torch.randn(b2.w_shape, device="cuda", generator=g) * 0.05).to(td)
    x = (torch.randn(b2.i_shape(N), device="cuda", generator=g) * 0.1).to(td)
    dy = (torch.randn(b2.o_shape(N), device="cuda", generator=g) * 0.1).to(td)
    y2 = b2.fprop(x, w); k2 = lib.last_kernel()
    torch.cuda.synchronize()
    y4 = b4.fprop(x, w); k4 = lib.last_kernel()
    torch.cuda.synchronize()
    d2 = b2.bprop(dy, w); d4 = b4.bprop(dy, w)
    torch.cuda.synchronize()
    e1, e2 = torch.equal(y2, y4), torch.equal(d2, d4)
    print("   kernels %d / %d  fprop identical %s  bprop identical %s  (max |diff| %.3e / %.3e)" %
          (k2, k4, e1, e2, (y2.float() - y4.float()).abs().max().item(), (d2.float() - d4.float()).abs().max().item()), flush=True)
    # the scheduled step order: the same sums in another order -> equal up to fp32 summation order (a 16-bit ulp here and there)
    b5 = balanced(lay)
    y5, d5 = b5.fprop(x, w), b5.bprop(dy, w)
    torch.cuda.synchronize()
    r1 = ((y5.float() - y2.float()).norm() / y2.float().norm().clamp_min(1e-30)).item()
    r2 = ((d5.float() - d2.float()).norm() / d2.float().norm().clamp_min(1e-30)).item()
    print("   scheduled order: rel L2 diff to staged %.2e / %.2e, elements that differ %.4f %% / %.4f %%" %
          (r1, r2, 100.0 * (y5 != y2).float().mean().item(), 100.0 * (d5 != d2).float().mean().item()), flush=True)
    ok = ok and e1 and e2 and k4 == lib.K_XCOL32_FLOW and r1 < 1e-3 and r2 < 1e-3
print("ALL IDENTICAL" if ok else "MISMATCH", flush=True)

_x = torch.randn(8192, 8192, device="cuda"); _t = time.time()
while time.time() - _t < 0.7: _x @ _x          # boost clock first
del _x
for d in (0.1, 0.2, 0.5):
    lay = P.random_layout(128, 128, d, seed=1234)
    b2, b4 = pair(lay)
    N = 8192
    w = (torch.randn(b2.w_shape, device="cuda") * 0.01).bfloat16()
    x = (torch.randn(b2.i_shape(N), device="cuda") * 0.1).bfloat16()
    dy = (torch.randn(b2.o_shape(N), device="cuda") * 0.1).bfloat16()
    fl = 2.0 * b2.blocks * 1024 * N
    b5 = balanced(lay)
    r = [timeit(lambda: b2.fprop(x, w)), timeit(lambda: b4.fprop(x, w)), timeit(lambda: b2.bprop(dy, w)), timeit(lambda: b4.bprop(dy, w)),
         timeit(lambda: b5.fprop(x, w)), timeit(lambda: b5.bprop(dy, w))]
    same = torch.equal(b2.fprop(x, w), b4.fprop(x, w)) and torch.equal(b2.bprop(dy, w), b4.bprop(dy, w))
    print("%s d%.2f fprop staged %.1f us flow(natural order) %.1f us flow(scheduled) %.1f us | bprop staged %.1f us flow(natural) %.1f us flow(scheduled) %.1f us | scheduled %.0f / %.0f TF | natural identical %s" %
          (os.environ.get("TAG", ""), d, r[0], r[1], r[4], r[2], r[3], r[5], fl / r[4] / 1e6, fl / r[5] / 1e6, same), flush=True)
lib.set_kernel_variant(0)
