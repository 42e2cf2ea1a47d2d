"""host time of one fprop / bprop / updat call through the Python class (eager mode): cProfile over 3000 calls at N = 64"""
import sys, os, time, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul
b = BlocksparseMatMul(P.random_layout(128, 128, 0.2, 1234), block_size=32, feature_axis=int(os.environ.get("AXIS", "1")))
w = (torch.randn(b.w_shape, device="cuda") * 0.05).bfloat16()
x = (torch.randn(b.i_shape(64), device="cuda") * 0.1).bfloat16()
dy = (torch.randn(b.o_shape(64), device="cuda") * 0.1).bfloat16()
dw = torch.empty(b.w_shape, dtype=torch.bfloat16, device="cuda")
for name, fn in (("fprop", lambda: b.fprop(x, w)), ("bprop", lambda: b.bprop(dy, w)), ("updat", lambda: b.updat(x, dy, dw=dw))):
    for _ in range(200): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3000): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s: host %.2f us per call (issue loop), %.2f us per call incl. drain" % (name, (t1 - t0) / 3000 * 1e6, (t2 - t0) / 3000 * 1e6), flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(3000): b.fprop(x, w)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
print(s.getvalue()[:3500])
