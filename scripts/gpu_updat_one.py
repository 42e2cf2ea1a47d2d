"""time updat (bsize 32, bf16, N = 8192; AXIS=0|1 in the environment, default 1) for the library selected by BSMM_LIB: density x plan option"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib

def timeit(fn, reps=100, warm=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

tag = os.path.basename(os.environ.get("BSMM_LIB", "default"))
cases = sys.argv[1:] or ["d10:stream16", "d20:stream16", "d50:stream8"]
OPT = {"stream32": _lib.PLAN_STREAM_32, "stream16": _lib.PLAN_STREAM_16, "stream8": _lib.PLAN_STREAM_8, "win8": _lib.PLAN_WINDOW_8, "auto": 0,
       "s16x1": _lib.PLAN_STREAM_16 | (1 << _lib.PLAN_UPDAT_SETS_SHIFT), "s16x2": _lib.PLAN_STREAM_16 | (2 << _lib.PLAN_UPDAT_SETS_SHIFT),
       "s16x4": _lib.PLAN_STREAM_16 | (4 << _lib.PLAN_UPDAT_SETS_SHIFT), "s16x8": _lib.PLAN_STREAM_16 | (8 << _lib.PLAN_UPDAT_SETS_SHIFT),
       "s8x4": _lib.PLAN_STREAM_8 | (4 << _lib.PLAN_UPDAT_SETS_SHIFT), "s8x2": _lib.PLAN_STREAM_8 | (2 << _lib.PLAN_UPDAT_SETS_SHIFT)}
out = []
for c in cases:
    d, o = c.split(":")
    CBK, NN = int(os.environ.get("CB", "128")), int(os.environ.get("N", "8192"))
    lay = P.random_layout(CBK, CBK, float(d[1:]) / 100.0, 1234)
    b = BlocksparseMatMul(lay, block_size=32, feature_axis=int(os.environ.get("AXIS", "1")), plan_options=OPT[o])
    g = torch.Generator(device="cuda").manual_seed(1)
    x = (torch.randn(b.i_shape(NN), device="cuda", generator=g) * 0.1).bfloat16()
    dy = (torch.randn(b.o_shape(NN), device="cuda", generator=g) * 0.1).bfloat16()
    dw = torch.empty(b.w_shape, dtype=torch.bfloat16, device="cuda")
    b.updat(x, dy, dw=dw); k = _lib.last_kernel()
    us = timeit(lambda: b.updat(x, dy, dw=dw))
    out.append("%s %.1f (k%d)" % (c, us, k))
print("%-28s %s" % (tag, "  ".join(out)), flush=True)
