"""On-disk format for block-sparse weights + dense <-> sparse converters.

The reference pickles only ``(layout, bsize, axis, z_order, name)`` and rebuilds the tables
(/root/reference/blocksparse/matmul.py:76-80,161-162); the weights ``W[blocks, bs, bs]`` are only meaningful together
with the layout and the z-order flag (the block numbering depends on both).  A checkpoint therefore stores all of them:

    format_version  int
    layout_bits     uint8[ceil(CB*KB/8)]   np.packbits of the row-major 0/1 layout
    layout_shape    int64[2]               (CB, KB)
    block_size, feature_axis, z_order
    dtype           str                    "float32" | "float16" | "bfloat16"
    W               float32/float16 array, or uint16 bit patterns for bfloat16 (NumPy has no bf16)

Pure NumPy (+ torch only to convert tensors); usable without a GPU.
"""
import numpy as np

from .lut import build_tables

FORMAT_VERSION = 1


def block_coords(layout, z_order=True):
    """(blocks, 2) int32 array: (c, k) of weight block w, in the numbering the lookup tables use."""
    return build_tables(layout, z_order=z_order)["updat_lut"]


def to_dense(layout, W, z_order=True):
    """Wdense (C, K) from W[blocks, bs, bs]; zero where the layout has no block."""
    W = np.asarray(W)
    coords = block_coords(layout, z_order)
    bs = W.shape[1]
    CB, KB = np.asarray(layout).shape
    if W.shape != (len(coords), bs, bs):
        raise ValueError("W has shape %s, layout needs (%d, bs, bs)" % (W.shape, len(coords)))
    Wd = np.zeros((CB, bs, KB, bs), dtype=W.dtype)
    Wd[coords[:, 0], :, coords[:, 1], :] = W
    return Wd.reshape(CB * bs, KB * bs)


def from_dense(layout, Wd, block_size, z_order=True):
    """W[blocks, bs, bs] gathered from a dense (C, K) matrix (entries outside the layout are dropped)."""
    Wd = np.asarray(Wd)
    CB, KB = np.asarray(layout).shape
    bs = block_size
    if Wd.shape != (CB * bs, KB * bs):
        raise ValueError("dense matrix has shape %s, layout x block_size needs %s" % (Wd.shape, (CB * bs, KB * bs)))
    coords = block_coords(layout, z_order)
    return np.ascontiguousarray(Wd.reshape(CB, bs, KB, bs)[coords[:, 0], :, coords[:, 1], :])


def renumber(W, layout, z_order_from, z_order_to):
    """Re-order weight blocks between the two numberings of one layout (e.g. a checkpoint written with z_order=False)."""
    a = block_coords(layout, z_order_from)
    b = block_coords(layout, z_order_to)
    KB = np.asarray(layout).shape[1]
    pos = {int(c) * KB + int(k): i for i, (c, k) in enumerate(a)}
    idx = np.array([pos[int(c) * KB + int(k)] for c, k in b], dtype=np.int64)
    return np.asarray(W)[idx]


def _to_numpy(W):
    """(array, dtype name).  torch bf16 tensors are stored as their uint16 bit patterns."""
    try:
        import torch
        if isinstance(W, torch.Tensor):
            if W.dtype == torch.bfloat16:
                return W.detach().cpu().view(torch.int16).numpy().view(np.uint16), "bfloat16"
            return W.detach().cpu().numpy(), str(W.dtype).replace("torch.", "")
    except ImportError:  # pragma: no cover
        pass
    W = np.asarray(W)
    return W, str(W.dtype)


def save(path, bsmm, W):
    """Write layout + numbering + weights of one BlocksparseMatMul to ``path`` (.npz)."""
    arr, dtype = _to_numpy(W)
    if tuple(arr.shape) != tuple(bsmm.w_shape):
        raise ValueError("W has shape %s, expected %s" % (arr.shape, bsmm.w_shape))
    layout = np.asarray(bsmm.layout, dtype=bool)
    np.savez_compressed(path, format_version=FORMAT_VERSION, layout_bits=np.packbits(layout.reshape(-1)),
                        layout_shape=np.array(layout.shape, dtype=np.int64), block_size=bsmm.bsize,
                        feature_axis=bsmm.axis, z_order=int(bsmm.z_order), dtype=dtype, W=arr)


def load(path, device=None):
    """-> (BlocksparseMatMul, W).  W is a NumPy array, or a torch tensor on ``device`` when one is given."""
    from .matmul import BlocksparseMatMul
    z = np.load(path, allow_pickle=False)
    if int(z["format_version"]) != FORMAT_VERSION:
        raise ValueError("unsupported checkpoint version %d" % int(z["format_version"]))
    CB, KB = (int(v) for v in z["layout_shape"])
    layout = np.unpackbits(z["layout_bits"])[:CB * KB].reshape(CB, KB).astype(bool)
    bsmm = BlocksparseMatMul(layout, block_size=int(z["block_size"]), feature_axis=int(z["feature_axis"]),
                             z_order=bool(int(z["z_order"])))
    W, dtype = z["W"], str(z["dtype"])
    if tuple(W.shape) != tuple(bsmm.w_shape):
        raise ValueError("checkpoint weights do not match its layout")
    if device is None:
        return bsmm, W
    import torch
    if dtype == "bfloat16":
        t = torch.from_numpy(W.view(np.int16).copy()).view(torch.bfloat16)
    else:
        t = torch.from_numpy(W.copy())
    return bsmm, t.to(device)
