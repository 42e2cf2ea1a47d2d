"""Does running fprop / updat / bprop of one step on three HIP streams overlap anything?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul
b = BlocksparseMatMul(P.random_layout(128, 128, 0.2, seed=1234), block_size=32, feature_axis=1)
N = 8192
w = (torch.randn(b.w_shape, device="cuda") * 0.01).bfloat16()
x = (torch.randn(b.i_shape(N), device="cuda") * 0.1).bfloat16()
dy = (torch.randn(b.o_shape(N), device="cuda") * 0.1).bfloat16()
dw = torch.empty(b.w_shape, dtype=torch.bfloat16, device="cuda")
ss = [torch.cuda.Stream() for _ in range(3)]
def seq():
    b.fprop(x, w); b.updat(x, dy, dw=dw); b.bprop(dy, w)
def par():
    cur = torch.cuda.current_stream()
    for s in ss: s.wait_stream(cur)
    with torch.cuda.stream(ss[0]): b.fprop(x, w)
    with torch.cuda.stream(ss[1]): b.updat(x, dy, dw=dw)
    with torch.cuda.stream(ss[2]): b.bprop(dy, w)
    for s in ss: cur.wait_stream(s)
for name, fn in (("sequential", seq), ("3 streams", par), ("sequential", seq), ("3 streams", par)):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 50
    print("%-11s %.4f ms/step  %.1f TF" % (name, el * 1e3, 3 * 2.0 * b.blocks * 1024 * N / el / 1e12), flush=True)
