"""Row-owner weight-gradient kernel for bsize 16 on feature axis 0 ('BSU6' section, csrc/bsmm_updat16_rows.h) against the windowed kernel
(plan option PLAN_UPDAT16_WINDOWED) and the float64 oracle on a set of shapes, then timing at BASELINE configs[2] for several splits.
Run under `timeout`: every case prints before it runs.  TIME_ONLY=1 skips the parity cases."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import _parity as P
from oracle import bsmm_oracle as orc
from blocksparse_amd import BlocksparseMatMul, _lib as lib


def timeit(fn, reps=50):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


cases = [("4x4 dense N=512", np.ones((4, 4), dtype=np.int32), 512, "bf16", 0),
         ("40x24 30% N=1000 (ragged last chunk)", P.random_layout(40, 24, 0.3, seed=2), 1000, "bf16", 0),
         ("33x35 25% N=520", P.random_layout(33, 35, 0.25, seed=3), 520, "f16", 0),
         ("70x96 10% N=2048 split 4", P.random_layout(70, 96, 0.10, seed=4), 2048, "bf16", 4),
         ("70x96 10% N=2048 split 1 (direct store)", P.random_layout(70, 96, 0.10, seed=4), 2048, "bf16", 1),
         ("64x64 10% N=8 (one short chunk) split 1", P.random_layout(64, 64, 0.10, seed=5), 8, "bf16", 1),
         ("256x256 10% N=4096", P.random_layout(256, 256, 0.10, seed=1234), 4096, "bf16", 0),
         ("256x256 5% N=1024 split 2", P.random_layout(256, 256, 0.05, seed=6), 1024, "f16", 2)]
ok = True
if not os.environ.get("TIME_ONLY"):
    for name, lay, N, dt, split in cases:
        print("case", name, flush=True)
        b_new = BlocksparseMatMul(lay, block_size=16, feature_axis=0, updat_split=split)
        b_old = BlocksparseMatMul(lay, block_size=16, feature_axis=0, plan_options=lib.PLAN_UPDAT16_WINDOWED)
        W, X, E = P.make_inputs(b_new.w_shape, b_new.i_shape(N), b_new.o_shape(N), dt, seed=11)
        x, e = P.to_dev(X, dt, torch), P.to_dev(E, dt, torch)
        lib.set_kernel_variant(3)
        d_new = b_new.updat(x, e); k_new = lib.last_kernel()
        d_old = b_old.updat(x, e); k_old = lib.last_kernel()
        dw0 = torch.randn(b_new.w_shape, device="cuda").to(d_new.dtype)
        d_ab = b_new.updat(x, e, alpha=0.5, beta=2.0, dw=dw0.clone())
        d2 = b_new.updat([x, x], [e, e])
        lib.set_kernel_variant(0)
        torch.cuda.synchronize()
        t = orc.build_layout_luts(np.asarray(lay), 16)
        ref = orc.updat(t, X.astype(np.float64), E.astype(np.float64), 0)
        dt_bar = 4 * P.L2_BAR[dt] if dt == "bf16" else P.L2_BAR[dt]      # (against the UNROUNDED float64 sums: bf16 rounding of the output alone is 1.7e-3)
        l2n, _ = P.errors(P.to_host(d_new), ref)
        l2o, _ = P.errors(P.to_host(d_old), ref)
        l2ab, _ = P.errors(P.to_host(d_ab), 0.5 * ref + 2.0 * P.to_host(dw0).astype(np.float64))
        l22, _ = P.errors(P.to_host(d2), 2.0 * ref)
        good = k_new == lib.K_UPDAT16_ROWS and l2n <= dt_bar and l2ab <= dt_bar and l22 <= dt_bar and abs(l2n - l2o) < 1e-5
        print("   kernels %d / %d   L2 vs float64: rows %.2e  windowed %.2e  alpha/beta %.2e  two pairs %.2e   %s" %
              (k_new, k_old, l2n, l2o, l2ab, l22, "ok" if good else "MISMATCH"), flush=True)
        ok = ok and good
    print("ALL OK" if ok else "MISMATCH", flush=True)

_x = torch.randn(8192, 8192, device="cuda"); _t = time.time()
while time.time() - _t < 0.7: _x @ _x          # boost clock first
del _x
lay = P.random_layout(256, 256, 0.10, seed=1234)
N = int(os.environ.get("N", "8192"))
b_old = BlocksparseMatMul(lay, block_size=16, feature_axis=0, plan_options=lib.PLAN_UPDAT16_WINDOWED)
x = (torch.randn(b_old.i_shape(N), device="cuda") * 0.1).bfloat16()
e = (torch.randn(b_old.o_shape(N), device="cuda") * 0.1).bfloat16()
fl = 2.0 * b_old.blocks * 256 * N
t_old = timeit(lambda: b_old.updat(x, e)); k_old = lib.last_kernel()
print("%s 4096^2 bs16 10%% axis 0 N=%d: windowed %.1f us = %.0f TF (kernel %d)" % (os.environ.get("TAG", ""), N, t_old, fl / t_old / 1e6, k_old), flush=True)
for split in [int(v) for v in os.environ.get("SPLITS", "0,2,4,8,16").split(",")]:
    b_new = BlocksparseMatMul(lay, block_size=16, feature_axis=0, updat_split=split)
    t_new = timeit(lambda: b_new.updat(x, e)); k_new = lib.last_kernel()
    print("   rows kernel split %2d: %.1f us = %.0f TF (kernel %d)" % (split, t_new, fl / t_new / 1e6, k_new), flush=True)
