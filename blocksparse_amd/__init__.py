"""blocksparse_amd -- MI355X-native block-sparse matmul engine (drop-in for the BlocksparseMatMul hot path
of openai/blocksparse).  Host mirror of the reference operator interface over libbsmm_hip.so."""
from .lut import build_tables, z_order_2d, ceil_div  # noqa: F401
from .matmul import BlocksparseMatMul  # noqa: F401
from .transformer import BlocksparseTransformer  # noqa: F401
from .sparse_proj import SparseProj  # noqa: F401
from . import checkpoint  # noqa: F401

__version__ = "0.1.0"
