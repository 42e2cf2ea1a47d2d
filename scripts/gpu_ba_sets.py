import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
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
lay = P.ba_layout(128, 14, seed=1)
for sets in (0, 1, 2, 4, 8):
    for win in (0, _lib.PLAN_STREAM_8):
        try:
            b = BlocksparseMatMul(lay, block_size=32, feature_axis=1, plan_options=(sets << _lib.PLAN_UPDAT_SETS_SHIFT) | win)
            g = torch.Generator(device="cuda").manual_seed(1)
            x = (torch.randn(b.i_shape(8192), device="cuda", generator=g) * 0.1).bfloat16()
            dy = (torch.randn(b.o_shape(8192), device="cuda", generator=g) * 0.1).bfloat16()
            b.updat(x, dy)
            plan = b._tables_on(x.device).updat_plan.host
            print("sets", sets, "win", hex(win), "ws", plan[2], "items", plan[4], "nsets", plan[8], "k", _lib.last_kernel(), "%.1f us" % timeit(lambda: b.updat(x, dy)), flush=True)
        except Exception as e:
            print("sets", sets, "win", hex(win), "ERR", str(e)[:100])
