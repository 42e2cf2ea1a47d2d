"""a few calls of one attention operator at BASELINE configs[4] (OP = nt | fused | sm | nn; fp32 activations): the workload of counter passes"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from blocksparse_amd import BlocksparseTransformer
from oracle import bst_oracle as O
op = os.environ.get("OP", "nt")
bst = BlocksparseTransformer(O.local_strided_layout(128), block_size=32, heads=16, mask_callback=O.causal_mask_callback)
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.rand((4, 4096, 1024), device="cuda", generator=g) * 2 - 1
k = torch.rand((4, 4096, 1024), device="cuda", generator=g) * 2 - 1
mask = bst._table("mask", q.device)
w = bst._nt(q, k, torch.bfloat16)
p = bst._softmax_fwd(w, 0.125, mask, torch.bfloat16)
fn = {"nt": lambda: bst._nt(q, k, torch.bfloat16), "fused": lambda: bst._nt_softmax(q, k, 0.125, mask, torch.bfloat16),
      "sm": lambda: bst._softmax_fwd(w, 0.125, mask, torch.bfloat16), "nn": lambda: bst._xn(p, k, False)}[op]
for _ in range(int(os.environ.get("REPS", "6"))): fn()
torch.cuda.synchronize()
