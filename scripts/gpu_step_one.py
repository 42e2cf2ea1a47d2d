"""time the bench step (fprop, bprop, updat back to back) and its passes for the library selected by BSMM_LIB"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib
tag = os.path.basename(os.environ.get("BSMM_LIB", "default"))
for d in [int(a) for a in sys.argv[1:]] or [20]:
    lay = P.random_layout(128, 128, d / 100.0, 1234)
    b = BlocksparseMatMul(lay, block_size=32, feature_axis=1)
    g = torch.Generator(device="cuda").manual_seed(1)
    w = (torch.randn(b.w_shape, device="cuda", generator=g) * 0.05).bfloat16()
    x = (torch.randn(b.i_shape(8192), device="cuda", generator=g) * 0.1).bfloat16()
    dy = (torch.randn(b.o_shape(8192), device="cuda", generator=g) * 0.1).bfloat16()
    dw = torch.empty_like(w)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    tot = [0.0, 0.0, 0.0]
    for it in range(130):
        ev[0].record(); b.fprop(x, w); ev[1].record(); b.bprop(dy, w); ev[2].record(); b.updat(x, dy, dw=dw); ev[3].record()
        if it >= 30:
            torch.cuda.synchronize()
            for k in range(3): tot[k] += ev[k].elapsed_time(ev[k + 1])
    print("%-22s d%d step %.1f us: fprop %.1f bprop %.1f updat %.1f" % (tag, d, sum(tot) * 10, tot[0] * 10, tot[1] * 10, tot[2] * 10), flush=True)
