"""L2-relative errors of the attention ops (fp32 activations, bf16 scores) against the float64 oracle: bf16-piece path vs fp32 MFMA path."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import make_golden_bst as G
import test_bst_gpu as T
from blocksparse_amd import BlocksparseTransformer
lay = G.layouts()["causal_2heads"]
res = T._run_case(torch, BlocksparseTransformer, lay, 2, 32, 64, 2, G.causal_cb, 11, "f32", "bf16")
print(os.environ.get("BST_XN_SPLIT", "1"), os.environ.get("BST_NT_SPLIT", "1"), {k: "%.2e" % v for k, v in res.items()})
