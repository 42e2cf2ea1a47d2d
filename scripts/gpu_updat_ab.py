"""A/B of the bsize-32 axis-1 updat kernels at the bench shape: plan options x density (x split).  GPU box only."""
import sys, os, time
import numpy as np
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

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
opts = [("stream(auto)", 0, 0), ("stream16", _lib.PLAN_STREAM_16, 0), ("stream8", _lib.PLAN_STREAM_8, 0), ("win8", _lib.PLAN_WINDOW_8, 0), ("win16w", _lib.PLAN_WINDOW_16W, 0)]
layouts = [("d10", P.random_layout(128, 128, 0.1, 1234)), ("d20", P.random_layout(128, 128, 0.2, 1234)), ("d50", P.random_layout(128, 128, 0.5, 1234)),
           ("BA", P.ba_layout(128, 14, seed=1)), ("cfg3", P.random_layout(256, 256, 0.05, 1234))]
# keep the GPU warm
for name, lay in layouts:
    g = torch.Generator(device="cuda").manual_seed(1)
    for oname, opt, split in opts:
        if name == "d50" and oname in ("stream16", "win16w"): continue
        b = BlocksparseMatMul(lay, block_size=32, feature_axis=1, plan_options=opt, updat_split=split)
        n = N if name != "cfg3" else 4096
        x = (torch.randn(b.i_shape(n), device="cuda", generator=g) * 0.1).bfloat16()
        dy = (torch.randn(b.o_shape(n), device="cuda", generator=g) * 0.1).bfloat16()
        dw = torch.empty(b.w_shape, dtype=torch.bfloat16, device="cuda")
        us = timeit(lambda: b.updat(x, dy, dw=dw))
        k = _lib.last_kernel()
        print("%-5s %-13s blocks %5d N %5d  %7.1f us  %7.1f TF  kernel %d" % (name, oname, b.blocks, n, us, 2.0 * b.blocks * 1024 * n / us / 1e6, k), flush=True)
