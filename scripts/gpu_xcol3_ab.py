"""A/B of the staged xprop schedules (bsize 32, bf16, N = 8192): round-2 'BSX2' (PLAN_XCOL_R2) against 'BSX3' with its duty / column
policies; every variant must give the bits of the round-1 kernel (same accumulation order).  python scripts/gpu_xcol3_ab.py [dens ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib

REPS = int(os.environ.get('XP_REPS', '100'))
AXIS = int(os.environ.get("XP_AXIS", "1"))
NN = int(os.environ.get("XP_N", "8192"))
def timeit(fn, reps=None):
    reps = reps or REPS
    for _ in range(max(3, reps // 3)): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

D, PM = _lib.PLAN_XPROP_DUTY_SHIFT, _lib.PLAN_XPROP_PERM_SHIFT
VARIANTS = [("r1", _lib.PLAN_XCOL_UNSTAGED), ("r2", _lib.PLAN_XCOL_R2), ("v3", 0), ("v3-lpt", 1 << D), ("v3-id", 1 << PM), ("v3-lpt-id", (1 << D) | (1 << PM)),
            ("v3-ph1", 1 << 8), ("v3-ph2", 2 << 8)]
sel = os.environ.get("XP_VARIANTS")
if sel:
    VARIANTS = [v for v in VARIANTS if v[0] in sel.split(",") or v[0] == "r1"]
dens = [int(a) for a in sys.argv[1:]] or [10, 20, 50]
for d in dens:
    lay = P.random_layout(128, 128, d / 100.0, 1234)
    ref = None
    for name, opt in VARIANTS:
        b = BlocksparseMatMul(lay, block_size=32, feature_axis=AXIS, plan_options=opt)
        g = torch.Generator(device="cuda").manual_seed(1)
        w = (torch.randn(b.w_shape, device="cuda", generator=g) * 0.05).bfloat16()
        x = (torch.randn(b.i_shape(NN), device="cuda", generator=g) * 0.1).bfloat16()
        dy = (torch.randn(b.o_shape(NN), device="cuda", generator=g) * 0.1).bfloat16()
        y, dx = b.fprop(x, w), b.bprop(dy, w)
        kf = _lib.last_kernel()
        if ref is None: ref = (y, dx)
        same = "bits f=%s b=%s" % (torch.equal(ref[0], y), torch.equal(ref[1], dx))
        tf, tb = timeit(lambda: b.fprop(x, w)), timeit(lambda: b.bprop(dy, w))
        print(os.path.basename(os.environ.get("BSMM_LIB", "default")), "axis%d N%d d%-3d %-12s f %6.1f b %6.1f (k%d) %s" % (AXIS, NN, d, name, tf, tb, kf, same), flush=True)
