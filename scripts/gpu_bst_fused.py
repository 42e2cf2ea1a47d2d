"""BASELINE configs[4] (batch 4, 16 heads x 64, ctx 4096, bsize 32, local + strided causal): scores + softmax as two launches against the fused launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from blocksparse_amd import BlocksparseTransformer
from oracle import bst_oracle as O

def timeit(fn, reps=50, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

bst = BlocksparseTransformer(O.local_strided_layout(128), block_size=32, heads=16, mask_callback=O.causal_mask_callback)
g = torch.Generator(device="cuda").manual_seed(1)
for dt in (torch.float32, torch.bfloat16):
    q = (torch.rand((4, 4096, 1024), device="cuda", generator=g) * 2 - 1).to(dt)
    k = (torch.rand((4, 4096, 1024), device="cuda", generator=g) * 2 - 1).to(dt)
    mask = bst._table("mask", q.device)
    sd = torch.bfloat16
    w = bst._nt(q, k, sd)
    t_nt = timeit(lambda: bst._nt(q, k, sd))
    t_sm = timeit(lambda: bst._softmax_fwd(w, 0.125, mask, sd))
    t_two = timeit(lambda: bst._softmax_fwd(bst._nt(q, k, sd), 0.125, mask, sd))
    t_f = timeit(lambda: bst._nt_softmax(q, k, 0.125, mask, sd))
    print("%s: nt %.1f us, softmax %.1f us, the two back to back %.1f us, fused %.1f us" % (dt, t_nt, t_sm, t_two, t_f), flush=True)
