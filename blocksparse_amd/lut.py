"""Host-side layout -> lookup-table builder (NumPy, vectorised).

Produces bit-identical tables to the reference builder
(/root/reference/blocksparse/matmul.py:82-162 ctor, :172-270 ``xprop_lut``) in the int32 format the
device kernels consume (see include/bsmm.h and SURVEY.md A.1):

    xprop lut  int32[4*S + 2*B]   header i: (entry_offset/2, n_entries, out_block, lock_id)
                                  entry  j: (in_block, weight_block)
    updat lut  int32[B, 2]        row w = (c, k) of weight block w (z-ordered when z_order=True)

Differences from the reference implementation (not from its output):
  * entries are sorted explicitly (k-major for fprop, stable c-major for bprop) instead of relying on
    the ordering of ``scipy.sparse.find`` (SURVEY TRAP 3);
  * Morton codes and segment cuts are computed with array arithmetic: the cost is O(blocks) NumPy work
    plus a Python loop over block *columns* only, so 10^5..10^6-block layouts build in well under a second.
"""
import numpy as np

SEG_MAX = (1 << 63) - 1


def ceil_div(x, y):
    return -(-x // y)


def z_order_2d(x, y):
    """Morton code of (x, y): x bits on even positions, y bits on odd (utils.py:95-103).  Works on
    Python ints and on integer arrays."""
    x = np.asarray(x, dtype=np.uint64)
    y = np.asarray(y, dtype=np.uint64)

    def spread(v):
        v = v & np.uint64(0xFFFFFFFF)
        v = (v | (v << np.uint64(16))) & np.uint64(0x0000FFFF0000FFFF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x00FF00FF00FF00FF)
        v = (v | (v << np.uint64(4))) & np.uint64(0x0F0F0F0F0F0F0F0F)
        v = (v | (v << np.uint64(2))) & np.uint64(0x3333333333333333)
        v = (v | (v << np.uint64(1))) & np.uint64(0x5555555555555555)
        return v

    out = spread(x) | (spread(y) << np.uint64(1))
    return int(out) if out.ndim == 0 else out


def segment_policy(layout):
    """Segment length limits from the per-column block counts (matmul.py:94-105)."""
    counts = np.asarray(layout).astype(np.int64).sum(axis=0)
    hi = int(counts.max())
    lo = int(counts[counts > 0].min())
    max_seg = max(ceil_div(hi, 4), lo * 2) if hi / lo > 2.0 else SEG_MAX
    min_seg = max(ceil_div(max_seg, 4), 4)
    return max_seg, min_seg


def xprop_table(n_out, in_blk, out_blk, wid, max_seg, min_seg):
    """Build one xprop table from blocks already ordered by (out_blk ascending, then visiting order).

    Returns dict(lut, l2_lut, shared, l2_shared, segments, locks, cols) where ``cols`` is the list
    ``[(out_block, [(in_block, w), ...]), ...]`` of unsegmented columns (the reference's
    fprop_list/bprop_list)."""
    in_blk = np.asarray(in_blk, dtype=np.int64)
    out_blk = np.asarray(out_blk, dtype=np.int64)
    wid = np.asarray(wid, dtype=np.int64)
    B = len(wid)
    counts = np.bincount(out_blk, minlength=n_out).astype(np.int64)
    present = np.nonzero(counts)[0]
    empty = np.nonzero(counts == 0)[0]
    starts = np.zeros(n_out + 1, dtype=np.int64)
    np.cumsum(counts, out=starts[1:])

    # a column of n entries is cut after every max_seg entries as long as >= min_seg entries remain
    # (matmul.py:218); the tail keeps whatever is left.
    n = counts[present]
    if max_seg >= SEG_MAX:
        cuts = np.zeros(len(present), dtype=np.int64)
    else:
        cuts = np.where(n >= min_seg, (n - min_seg) // max_seg, 0)
    nseg = cuts + 1
    S = int(nseg.sum()) + len(empty)

    seg_out = np.concatenate([np.repeat(present, nseg), empty])
    seg_len = np.zeros(S, dtype=np.int64)
    seg_first = np.zeros(S, dtype=np.int64)       # index of first entry in the ordered block list
    lock = np.zeros(S, dtype=np.int64)
    pos = 0
    locks = 0
    for col, total, c in zip(present, n, cuts):
        base = starts[col]
        for t in range(c):
            seg_len[pos] = max_seg
            seg_first[pos] = base + t * max_seg
            pos += 1
        seg_len[pos] = total - c * max_seg
        seg_first[pos] = base + c * max_seg
        pos += 1
        if c > 0:
            locks += 1
            lock[pos - c - 1:pos] = locks
    # empty columns: length 0; their offset is wherever the entry cursor ended (= all entries consumed)
    seg_first[pos:] = B

    lut = np.empty(4 * S + 2 * B, dtype=np.int32)
    hdr = lut[:4 * S].reshape(S, 4)
    hdr[:, 0] = (4 * S + 2 * seg_first) // 2
    hdr[:, 1] = seg_len
    hdr[:, 2] = seg_out
    hdr[:, 3] = lock
    ent = lut[4 * S:].reshape(B, 2)
    ent[:, 0] = in_blk
    ent[:, 1] = wid

    # weight-norm table: unsegmented columns, one int32 (w) per entry, even total length (:254-268)
    Cn = len(present) + len(empty)
    size = 4 * Cn + B
    size += size & 1
    l2 = np.zeros(size, dtype=np.int32)
    l2h = l2[:4 * Cn].reshape(Cn, 4)
    col_order = np.concatenate([present, empty])
    l2h[:, 0] = 4 * Cn + np.concatenate([starts[present], np.full(len(empty), B, dtype=np.int64)])
    l2h[:, 1] = counts[col_order]
    l2h[:, 2] = col_order
    l2[4 * Cn:4 * Cn + B] = wid

    cols = []
    il = in_blk.tolist()
    wl = wid.tolist()
    for col in present.tolist():
        a, b = int(starts[col]), int(starts[col + 1])
        cols.append((col, list(zip(il[a:b], wl[a:b]))))
    for col in empty.tolist():
        cols.append((col, []))

    return dict(lut=lut, l2_lut=l2, shared=int(seg_len.max(initial=0)) * 8,
                l2_shared=int(counts.max(initial=0)) * 4, segments=S, locks=locks, cols=cols)


def build_tables(layout, z_order=True, segmented=True):
    """All host tables for a 0/1 block layout of shape (CB, KB).

    ``segmented=False`` applies the "not worth segmenting" branch of the reference policy to every
    layout: one segment per output block, hence no locks and a deterministic, single-writer kernel."""
    layout = np.asarray(layout) != 0
    if layout.ndim != 2:
        raise ValueError("layout must be 2-D (CB, KB)")
    CB, KB = layout.shape
    if not layout.any():
        raise ValueError("layout has no blocks")
    max_seg, min_seg = segment_policy(layout)
    if not segmented:
        max_seg, min_seg = SEG_MAX, max(ceil_div(SEG_MAX, 4), 4)

    cs, ks = np.nonzero(layout)
    o = np.lexsort((cs, ks))                 # k-major, c-minor: the fprop visiting order
    cs = cs[o].astype(np.int64)
    ks = ks[o].astype(np.int64)
    B = len(cs)

    wid = np.empty(B, dtype=np.int64)
    if z_order:
        zo = np.argsort(z_order_2d(cs, ks), kind="stable")
        wid[zo] = np.arange(B)
        updat = np.stack([cs[zo], ks[zo]], axis=1)
    else:
        wid[:] = np.arange(B)
        updat = np.stack([cs, ks], axis=1)

    t = np.argsort(cs, kind="stable")        # c-major, k-minor: the bprop visiting order
    f = xprop_table(KB, cs, ks, wid, max_seg, min_seg)
    b = xprop_table(CB, ks[t], cs[t], wid[t], max_seg, min_seg)
    return dict(CB=CB, KB=KB, blocks=B, layout=layout, updat_lut=np.ascontiguousarray(updat, dtype=np.int32),
                fprop=f, bprop=b)


def double_tables(tables):
    """The tables of a gated call that runs on the UNGATED kernels over two weight images (include/bsmm.h, bsmm_gate_weights): every
    xprop entry (c, w) is followed by (c, w + blocks) -- image 0 holds round(g w), image 1 round(g w - image 0) -- so a column sums
    hi and lo of each of its blocks back to back; ``blocks`` doubles, headers keep their order, segment lengths and offsets double.
    The updat table is repeated only to keep the dict complete (a doubled table is never used for a weight gradient)."""
    B = int(tables["blocks"])
    out = dict(tables)
    out["blocks"] = 2 * B
    out["updat_lut"] = np.concatenate([tables["updat_lut"], tables["updat_lut"]], axis=0)
    for side in ("fprop", "bprop"):
        t = tables[side]
        S = int(t["segments"])
        lut = np.asarray(t["lut"], dtype=np.int32)
        hdr = lut[:4 * S].reshape(S, 4)
        ent = lut[4 * S:].reshape(B, 2)
        new = np.empty(4 * S + 4 * B, dtype=np.int32)
        nh = new[:4 * S].reshape(S, 4)
        nh[:] = hdr
        nh[:, 0] = 2 * S + 2 * (hdr[:, 0] - 2 * S)
        nh[:, 1] = 2 * hdr[:, 1]
        ne = new[4 * S:].reshape(B, 2, 2)
        ne[:, 0, :] = ent
        ne[:, 1, 0] = ent[:, 0]
        ne[:, 1, 1] = ent[:, 1] + B
        d = dict(t)
        d["lut"] = new
        d["shared"] = 2 * int(t["shared"])
        d["cols"] = [(col, [e for (c, w) in lst for e in ((c, w), (c, w + B))]) for col, lst in t["cols"]]
        out[side] = d
    return out
