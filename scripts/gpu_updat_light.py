"""streaming updat, light windows / small overflow pieces as direct blocks (default) against BSMM_PLAN_UPDAT_NO_DIRECT: time and bit identity, bf16, N = 8192"""
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
pl = (rs.rand(128, 128) < np.minimum(1.0, 6.0 / (1 + np.arange(128))[None, :])).astype(np.int32)
cases = [("BA(128, 14) + I", P.ba_layout(128, 14, seed=1)), ("power-law columns", np.maximum(pl, np.eye(128, dtype=np.int32))),
         ("uniform 20 %", P.random_layout(128, 128, 0.2, 1234)), ("uniform 21.5 %", P.random_layout(128, 128, 0.215, 5)), ("uniform 10 %", P.random_layout(128, 128, 0.1, 1234))]
for name, lay in cases:
    outs, line = {}, []
    for tag, opt in (("no direct", _lib.PLAN_UPDAT_NO_DIRECT), ("default", 0)):
        b = BlocksparseMatMul(lay, block_size=32, feature_axis=1, plan_options=opt)
        g = torch.Generator(device="cuda").manual_seed(1)
        x = (torch.randn(b.i_shape(8192), device="cuda", generator=g) * 0.1).bfloat16()
        dy = (torch.randn(b.o_shape(8192), device="cuda", generator=g) * 0.1).bfloat16()
        outs[tag] = b.updat(x, dy)
        assert _lib.last_kernel() == _lib.K_UPDAT_STREAM
        plan = b._tables_on(x.device).updat_plan.host
        line.append("%s (%d items, %d direct) %.1f" % (tag, plan[4], plan[28], timeit(lambda: b.updat(x, dy))))
    d = (outs["no direct"].float() - outs["default"].float()).abs().max().item()
    print("%s (%d blocks): %s | max |diff| %.3g" % (name, int(lay.sum()), " | ".join(line), d), flush=True)
