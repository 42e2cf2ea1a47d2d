"""Does the activation row pitch matter (L2 channel camping)?  Same blocks-per-column statistics with C = 4096 (pitch 8192 B), 4160 (8320 B)
and 4224 features: fprop reads X (pitch C), bprop reads DY (pitch K = 4096 always).  us per call and per 1000 blocks."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib
def timeit(fn, reps=60):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
opt = int(os.environ.get("XP_OPT", "0"), 0)
for CB, KB in ((128, 128), (130, 128), (132, 128), (136, 128), (128, 130), (144, 128)):
    lay = P.random_layout(CB, KB, 0.2, 1234)
    b = BlocksparseMatMul(lay, block_size=32, feature_axis=1, plan_options=opt)
    g = torch.Generator(device="cuda").manual_seed(1)
    w = (torch.randn(b.w_shape, device="cuda", generator=g) * 0.05).bfloat16()
    x = (torch.randn(b.i_shape(8192), device="cuda", generator=g) * 0.1).bfloat16()
    dy = (torch.randn(b.o_shape(8192), device="cuda", generator=g) * 0.1).bfloat16()
    tf, tb = timeit(lambda: b.fprop(x, w)), timeit(lambda: b.bprop(dy, w))
    nb = int(lay.sum())
    print("CB %d KB %d blocks %d opt %#x: fprop %.1f us (%.2f per 100 blocks; X pitch %d B)  bprop %.1f us (%.2f; DY pitch %d B)" % (
        CB, KB, nb, opt, tf, tf / nb * 100, CB * 64, tb, tb / nb * 100, KB * 64), flush=True)
