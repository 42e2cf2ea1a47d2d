"""time fprop / bprop / updat at 4096x4096, bsize 8, 10 %, bf16, N = 8192 (both axes) for the library selected by BSMM_LIB"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib

def timeit(fn, reps=50, warm=15):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

dens = float(os.environ.get("DENS", "0.1"))
lay = P.random_layout(512, 512, dens, 1234)
for axis in (1, 0):
    b = BlocksparseMatMul(lay, block_size=8, feature_axis=axis)
    g = torch.Generator(device="cuda").manual_seed(1)
    w = (torch.randn(b.w_shape, device="cuda", generator=g) * 0.05).bfloat16()
    x = (torch.randn(b.i_shape(8192), device="cuda", generator=g) * 0.1).bfloat16()
    dy = (torch.randn(b.o_shape(8192), device="cuda", generator=g) * 0.1).bfloat16()
    dw = torch.empty_like(w)
    b.fprop(x, w); kf = _lib.last_kernel(); b.bprop(dy, w); b.updat(x, dy, dw=dw); ku = _lib.last_kernel()
    tf, tb, tu = timeit(lambda: b.fprop(x, w)), timeit(lambda: b.bprop(dy, w)), timeit(lambda: b.updat(x, dy, dw=dw))
    fl = 2.0 * b.blocks * 64 * 8192
    print("bs8 d%.0f%% axis %d blocks %d: fprop %.1f us (k%d, %.0f TF)  bprop %.1f (%.0f TF)  updat %.1f (k%d, %.0f TF)" %
          (dens * 100, axis, b.blocks, tf, kf, fl / tf / 1e6, tb, fl / tb / 1e6, tu, ku, fl / tu / 1e6), flush=True)
