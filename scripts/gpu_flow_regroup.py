"""'BSX4' plans version 4: regrouped output blocks (default on unbalanced layouts) against groups of 16 consecutive ones (BSMM_PLAN_FLOW_CONSECUTIVE):
bit identity and time of fprop / bprop at N = 8192, bf16, feature axis 1 -- the reference's Barabasi-Albert bench layout, a power-law layout, uniform 20 %"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib

def timeit(fn, reps=60, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

rs = np.random.RandomState(7)
pl = (rs.rand(128, 128) < np.minimum(1.0, 6.0 / (1 + np.arange(128))[None, :])).astype(np.int32)      # column k holds ~ 6 / (k + 1) of the rows
cases = [("BA(128, 14) + I", P.ba_layout(128, 14, seed=1)), ("power-law columns", np.maximum(pl, np.eye(128, dtype=np.int32))), ("uniform 20 %", P.random_layout(128, 128, 0.2, 1234))]
for name, lay in cases:
    outs = {}
    line = []
    for tag, opt in (("consecutive", _lib.PLAN_FLOW_CONSECUTIVE), ("default", 0)):
        b = BlocksparseMatMul(lay, block_size=32, feature_axis=1, plan_options=opt)
        g = torch.Generator(device="cuda").manual_seed(1)
        w = (torch.randn(b.w_shape, device="cuda", generator=g) * 0.05).bfloat16()
        x = (torch.randn(b.i_shape(8192), device="cuda", generator=g) * 0.1).bfloat16()
        dy = (torch.randn(b.o_shape(8192), device="cuda", generator=g) * 0.1).bfloat16()
        y, dx = b.fprop(x, w), b.bprop(dy, w)
        assert _lib.last_kernel() == _lib.K_XCOL32_FLOW
        outs[tag] = (y, dx)
        line.append("%s f %.1f b %.1f" % (tag, timeit(lambda: b.fprop(x, w)), timeit(lambda: b.bprop(dy, w))))
    same = torch.equal(outs["consecutive"][0], outs["default"][0]) and torch.equal(outs["consecutive"][1], outs["default"][1])
    print("%s (%d blocks): %s | bit-identical %s" % (name, int(lay.sum()), " | ".join(line), same), flush=True)
