"""BASELINE configs[4]: block-sparse attention, batch 4, heads 16 x 64, ctx 4096, bsize 32, local+strided causal layout.
Times every op and prints it next to its bound (algorithmic bytes at 8 TB/s, flops at the fp32 MFMA peak 157.3 TF)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from blocksparse_amd import BlocksparseTransformer

def timeit(fn, reps=100):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

q_, k_ = np.indices((128, 128))
lay = ((k_ <= q_) & ((q_ - k_ < 4) | ((q_ - k_) % 8 == 0))).astype(np.int32)
def cb(shape, h, q, k, b):
    m = np.ones(shape, dtype=bool)
    return np.tril(m) if q == k else m
act = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[sys.argv[1] if len(sys.argv) > 1 else "f32"]
B, H, HS, BS = 4, 16, 64, 32
bst = BlocksparseTransformer(lay, block_size=BS, heads=H, mask_callback=cb)
sd = torch.float16 if act == torch.float16 else torch.bfloat16
q = (torch.rand(B, 4096, H * HS, device="cuda") * 2 - 1).to(act)
k = (torch.rand(B, 4096, H * HS, device="cuda") * 2 - 1).to(act)
v = (torch.rand(B, 4096, H * HS, device="cuda") * 2 - 1).to(act)
w = bst._nt(q, k, sd)
a = bst._softmax_fwd(w, 0.125, bst._table("mask", "cuda"), sd)
da = torch.randn_like(a)
esz = q.element_size()
flops = 2.0 * B * H * bst.blocks * BS * BS * HS
sbytes = B * H * bst.blocks * BS * BS * 2
abytes = q.numel() * esz
rows = [
    ("nt", lambda: bst._nt(q, k, sd), flops, 2 * abytes + sbytes),
    ("softmax", lambda: bst._softmax_fwd(w, 0.125, bst._table("mask", "cuda"), sd), 0, 2 * sbytes),
    ("softmax_nomask", lambda: bst._softmax_fwd(w, 0.125, None, sd), 0, 2 * sbytes),
    ("nn", lambda: bst._xn(a, v, False), flops, 2 * abytes + sbytes),
    ("tn", lambda: bst._xn(a, q, True), flops, 2 * abytes + sbytes),
    ("softmax_grad", lambda: bst._softmax_bwd(da, a, 0.125), 0, 3 * sbytes),
]
peak = 157.3e12 if act == torch.float32 else 2500e12      # fp32 activations: f32 MFMA; 16-bit: v_mfma_f32_32x32x16
tot = 0.0
for name, fn, fl, by in rows:
    t = timeit(fn)
    tot += t
    bound = max(fl / peak, by / 8e12) * 1e3
    print("%-13s %8.3f ms | %7.1f TF  %7.1f GB/s | bound %.3f ms (%s) -> %4.1f%%" % (
        name, t, fl / t / 1e9, by / t / 1e6, bound, "mfma" if fl / peak > by / 8e12 else "hbm", 100 * bound / t), flush=True)
print("forward (nt + softmax + nn): see rows; sum of all five %.3f ms" % tot)
