"""Row-split kernel ('BSX5', bsmm_xrows.h) against the flow kernel ('BSX4'): bit identity on a set of shapes (with a per-tile / per-column
map of the differences when there are any), then timing at the bench workload.  Run under `timeout`: every case prints before it runs.
TIME_ONLY=1 skips the identity cases (variant builds); TAG names the build in the timing lines."""
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
    b4 = BlocksparseMatMul(layout, block_size=32, feature_axis=1); b4.rows = False
    b5 = BlocksparseMatMul(layout, block_size=32, feature_axis=1); b5.rows = True
    return b4, b5


def diff_map(a, b, name):
    """where two (N, F) outputs differ: 128-row tiles x 32-feature blocks"""
    d = (a.float() - b.float()).abs()
    N, F = d.shape
    nt, nb = (N + 127) // 128, F // 32
    pad = torch.zeros(nt * 128, F, device=d.device); pad[:N] = d
    m = pad.view(nt, 128, nb, 32).amax(dim=(1, 3)).cpu().numpy()
    bad = np.argwhere(m > 0)
    print("      %s: %d of %d (tile, block) cells differ; first: %s; max %.3e; rows-in-tile of first cell: %s" %
          (name, len(bad), m.size, bad[:8].tolist(), m.max(),
           (pad.view(nt, 128, nb, 32)[bad[0][0], :, bad[0][1], :].amax(dim=1) > 0).nonzero().flatten()[:16].tolist() if len(bad) else []), flush=True)


cases = [("tiny 4x4 N=128", P.random_layout(4, 4, 0.6, seed=1), 128, torch.bfloat16),
         ("16x16 dense N=256", np.ones((16, 16), dtype=np.int32), 256, torch.bfloat16),
         ("40x24 N=1000 (ragged rows, partial group)", P.random_layout(40, 24, 0.3, seed=2), 1000, torch.bfloat16),
         ("33x35 N=520 (odd block counts)", P.random_layout(33, 35, 0.25, seed=3), 520, torch.float16),
         ("128x128 20% N=2048", P.random_layout(128, 128, 0.2, seed=1234), 2048, torch.bfloat16),
         ("128x128 55% N=1024 (split steps)", P.random_layout(128, 128, 0.55, seed=5), 1024, torch.bfloat16),
         ("64x33 dense N=384 (three steps per pair)", np.ones((64, 33), dtype=np.int32), 384, torch.bfloat16),
         ("300x16 5% N=640 (> 64 steps)", P.random_layout(300, 16, 0.05, seed=6), 640, torch.bfloat16),
         ("BA 128 N=4096", P.ba_layout(128, 14, seed=1), 4096, torch.bfloat16),
         ("15x33 with an output group without blocks, N=520", np.eye(15, 33, dtype=np.int32), 520, torch.float16),
         ("128x128 20% N=8192 (bench shape: two units per CU)", P.random_layout(128, 128, 0.2, seed=1234), 8192, torch.bfloat16),
         ("128x128 20% N=12288 (three units per CU)", P.random_layout(128, 128, 0.2, seed=4321), 12288, torch.bfloat16)]
lib.set_kernel_variant(3)
ok = True
if not os.environ.get("TIME_ONLY"):
    for name, lay, N, td in cases:
        print("case", name, flush=True)
        b4, b5 = pair(lay)
        g = torch.Generator(device="cuda").manual_seed(7)
        w = (torch.randn(b4.w_shape, device="cuda", generator=g) * 0.05).to(td)
        x = (torch.randn(b4.i_shape(N), device="cuda", generator=g) * 0.1).to(td)
        dy = (torch.randn(b4.o_shape(N), device="cuda", generator=g) * 0.1).to(td)
        # (the host class picks the rows plan only when the units fill the chip: attach it here whatever the size)
        tabs = b5._tables_on(x.device)
        assert tabs.fprop_rows is not None and tabs.bprop_rows is not None, "no BSX5 plan for this layout"
        b5._xprop_plan = lambda tabs_, which, N_, nf, dtype, gate: getattr(tabs_, which + "_rows")
        y4 = b4.fprop(x, w); k4 = lib.last_kernel()
        torch.cuda.synchronize()
        y5 = b5.fprop(x, w); k5 = lib.last_kernel()
        torch.cuda.synchronize()
        d4 = b4.bprop(dy, w); d5 = b5.bprop(dy, w)
        torch.cuda.synchronize()
        # twice more: a race shows as run-to-run differences
        y5b, d5b = b5.fprop(x, w), b5.bprop(dy, w)
        torch.cuda.synchronize()
        e1, e2 = torch.equal(y4, y5), torch.equal(d4, d5)
        st = torch.equal(y5, y5b) and torch.equal(d5, d5b)
        print("   kernels %d / %d  fprop identical %s  bprop identical %s  stable %s (max |diff| %.3e / %.3e)" %
              (k4, k5, e1, e2, st, (y4.float() - y5.float()).abs().max().item(), (d4.float() - d5.float()).abs().max().item()), flush=True)
        if not e1: diff_map(y4, y5, "fprop")
        if not e2: diff_map(d4, d5, "bprop")
        ok = ok and e1 and e2 and st and k5 == lib.K_XCOL32_ROWS
    print("ALL IDENTICAL" if ok else "MISMATCH", flush=True)

_x = torch.randn(8192, 8192, device="cuda"); _t = time.time()
while time.time() - _t < 0.7: _x @ _x          # boost clock first
del _x
for d in [float(v) for v in os.environ.get("DENS", "0.1,0.2,0.5").split(",")]:
    lay = P.random_layout(128, 128, d, seed=1234)
    b4, b5 = pair(lay)
    N = 8192
    w = (torch.randn(b4.w_shape, device="cuda") * 0.01).bfloat16()
    x = (torch.randn(b4.i_shape(N), device="cuda") * 0.1).bfloat16()
    dy = (torch.randn(b4.o_shape(N), device="cuda") * 0.1).bfloat16()
    fl = 2.0 * b4.blocks * 1024 * N
    r = [timeit(lambda: b4.fprop(x, w)), timeit(lambda: b5.fprop(x, w)), timeit(lambda: b4.bprop(dy, w)), timeit(lambda: b5.bprop(dy, w))]
    k5 = lib.last_kernel()
    same = torch.equal(b4.fprop(x, w), b5.fprop(x, w)) and torch.equal(b4.bprop(dy, w), b5.bprop(dy, w))
    print("%s d%.2f fprop flow %.1f us rows %.1f us | bprop flow %.1f us rows %.1f us | rows %.0f / %.0f TF (kernel %d) | identical %s" %
          (os.environ.get("TAG", ""), d, r[0], r[1], r[2], r[3], fl / r[1] / 1e6, fl / r[3] / 1e6, k5, same), flush=True)
lib.set_kernel_variant(0)
