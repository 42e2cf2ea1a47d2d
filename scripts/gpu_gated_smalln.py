"""gated fprop at short minibatches: the GATED kernels (gate_images = False) against gated weight images + the ungated call, graph-free eager timing"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib

def timeit(fn, reps=200, warm=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for bs, axis in ((32, 1), (32, 0), (16, 0), (16, 1)):
    lay = P.random_layout(128, 128, 0.2, 1234) if bs == 32 else P.random_layout(256, 256, 0.1, 1234)
    b = BlocksparseMatMul(lay, block_size=bs, feature_axis=axis)
    b.GATE_IMAGES_MIN_N = {32: 0, 16: 0}
    g = torch.Generator(device="cuda").manual_seed(1)
    w = (torch.randn(b.w_shape, device="cuda", generator=g) * 0.05).bfloat16()
    gate = torch.rand(b.blocks, device="cuda", generator=g) * 2 - 0.5
    gate[::7] = 0
    mask = (torch.rand(b.blocks, device="cuda", generator=g) < 0.8).float()
    for N in (64, 128, 256, 512, 1024, 2048):
        x = (torch.randn(b.i_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
        out = []
        for name, gt, img in (("ungated", None, True), ("gated in-kernel", gate, False), ("gated images", gate, True), ("mask in-kernel", mask, False), ("mask image", mask, True)):
            b.gate_images = img
            b.fprop(x, w, gate=gt); k = _lib.last_kernel()
            out.append("%s %.1f (k%d)" % (name, timeit(lambda: b.fprop(x, w, gate=gt)), k))
        print("bsize %d axis %d N %d: %s" % (bs, axis, N, " | ".join(out)), flush=True)
