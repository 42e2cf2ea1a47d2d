"""time fprop / bprop (bsize 32, axis 1, bf16, N = 8192) for the library selected by BSMM_LIB: staged ('BSX2') vs default
plans per density, and check that both give the same bits (same MFMA order)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib

REPS = int(os.environ.get('XP_REPS', '100'))
def timeit(fn, reps=None, warm=None):
    reps = reps or REPS; warm = warm or max(3, REPS // 3)
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

tag = os.path.basename(os.environ.get("BSMM_LIB", "default"))
dens = [int(a) for a in sys.argv[1:]] or [10, 20, 50]
check = os.environ.get("XP_CHECK", "1") == "1"
for d in dens:
    CB = int(os.environ.get("XP_CB", "128"))
    lay = P.random_layout(CB, CB, d / 100.0, 1234)
    res = {}
    for name, opt in (("base", _lib.PLAN_XCOL_UNSTAGED), ("staged", int(os.environ.get("XP_OPT", "0"), 0))):
        b = BlocksparseMatMul(lay, block_size=32, feature_axis=int(os.environ.get("XP_AXIS", "1")), plan_options=opt)
        g = torch.Generator(device="cuda").manual_seed(1)
        w = (torch.randn(b.w_shape, device="cuda", generator=g) * 0.05).bfloat16()
        x = (torch.randn(b.i_shape(8192), device="cuda", generator=g) * 0.1).bfloat16()
        dy = (torch.randn(b.o_shape(8192), device="cuda", generator=g) * 0.1).bfloat16()
        y, dx = b.fprop(x, w), b.bprop(dy, w)
        kf = _lib.last_kernel()
        res[name] = (timeit(lambda: b.fprop(x, w)), timeit(lambda: b.bprop(dy, w)), y, dx, kf)
    same = ""
    if check:
        same = " same_bits fprop=%s bprop=%s" % (torch.equal(res["base"][2], res["staged"][2]), torch.equal(res["base"][3], res["staged"][3]))
    print("%-24s cb%d d%-3d base f %.1f b %.1f (k%d) | staged f %.1f b %.1f (k%d)%s" % (tag, CB, d, res["base"][0], res["base"][1], res["base"][4],
          res["staged"][0], res["staged"][1], res["staged"][4], same), flush=True)
