"""One-shot GPU diagnostic: parity table over every (dtype, bsize, axis) + timing sweep.  Prints, never asserts,
so a single gpurun call shows the state of every kernel."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib


def parity_table():
    layout = P.ba_layout(24, 3, seed=2)
    print("== parity (L2-rel vs oracle; BA(24,3) layout) ==")
    for variant in (0, 1):
        _lib.load().bsmm_set_kernel_variant(variant)
        for dtype in ("f32", "bf16", "f16"):
            for bs in (32, 16, 8):
                for axis in (0, 1):
                    for N in (64, 100):
                        try:
                            r = P.run_case(torch, BlocksparseMatMul, layout, bs, axis, dtype, N, seed=1)
                            s = " ".join("%s=%.1e" % (k, v[0]) for k, v in r.items())
                            bad = any(v[0] > P.L2_BAR[dtype] for v in r.values())
                        except Exception as ex:  # noqa
                            s, bad = "EXC %r" % (ex,), True
                        print("variant%d %-4s bs%-2d a%d N%-4d %s %s" % (variant, dtype, bs, axis, N, s, "<<<<< FAIL" if bad else ""))
    _lib.load().bsmm_set_kernel_variant(0)


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def timing():
    print("== timing: 4096x4096, effective TFLOP/s per pass ==")
    for dtype, td in (("bf16", torch.bfloat16), ("f32", torch.float32)):
        for bs, dens in ((32, 0.2), (32, 0.1), (32, 0.5), (16, 0.1), (8, 0.1)):
            CB = 4096 // bs
            layout = P.random_layout(CB, CB, dens, seed=1234)
            for axis in (1, 0):
                b = BlocksparseMatMul(layout, block_size=bs, feature_axis=axis)
                for N in (512, 8192):
                    if dtype == "f32" and (bs != 32 or dens != 0.2):
                        continue
                    if bs == 8 and N > 512:
                        continue
                    w = (torch.randn(b.w_shape, device="cuda") * 0.01).to(td)
                    x = (torch.randn(b.i_shape(N), device="cuda") * 0.1).to(td)
                    dy = (torch.randn(b.o_shape(N), device="cuda") * 0.1).to(td)
                    fl = 2.0 * b.blocks * bs * bs * N
                    reps = 10 if N >= 8192 else 20
                    tf = timeit(lambda: b.fprop(x, w), reps)
                    tb = timeit(lambda: b.bprop(dy, w), reps)
                    tu = timeit(lambda: b.updat(x, dy), reps)
                    print("%-4s bs%-2d d%.2f a%d N%-5d blocks %-6d  fprop %8.3f ms %7.1f TF | bprop %8.3f ms %7.1f TF | updat %8.3f ms %7.1f TF"
                          % (dtype, bs, dens, axis, N, b.blocks, tf, fl / tf / 1e9, tb, fl / tb / 1e9, tu, fl / tu / 1e9), flush=True)


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), "cpus", os.cpu_count())
    t0 = time.time()
    if "notable" not in sys.argv:
        parity_table()
    print("parity took %.1fs" % (time.time() - t0))
    if "notime" not in sys.argv:
        timing()
