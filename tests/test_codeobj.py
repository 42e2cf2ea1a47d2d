"""Build gate on register spills: the per-kernel metadata of the gfx950 code objects inside the BUILT library
(.vgpr_spill_count / .sgpr_spill_count / .private_segment_fixed_size, tests/_codeobj.py) must be zero for every kernel of the library --
a one-line edit once pushed xcol32sf_kernel (BASELINE configs[1]) from 127 registers to 128 + 8 spilled and cost a fifth of its
throughput without any test noticing.  The few kernels that still spill are listed by name with the count they may NOT exceed; nothing
on the hot list may appear there."""
import os
import re

import pytest

import _codeobj as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "blocksparse_amd", "libbsmm_hip.so")

# kernels of the default dispatch at the BASELINE configs and the north-star sweep: never a spill, never scratch
HOT = ["xflow32_kernel", "updat32_a1_v2_kernel", "updat2_reduce_kernel", "updat16_rows_kernel", "updat16_rows_finalize_kernel", "xcol16_list_kernel",
       "xcol32sf_kernel", "xcol32s_kernel", "split3_w_kernel", "split3_x_kernel", "xmid32_kernel", "xsmall32_kernel", "updat32_a1_small_kernel",
       "xcol32_v2_kernel", "updat16_win_kernel", "updat_finalize_kernel", "bst_nt_mfma_kernel", "bst_nt_mfma_direct_kernel", "bst_xn_mfma_kernel",
       "bst_xn_mfma16_kernel", "bst_xn_split_kernel", "bst_softmax_grad_kernel"]
# known spillers: (kernel name, substring of the mangled template arguments or None) -> (max vgpr spills, max sgpr spills).  They may shrink, not grow.
KNOWN = {
    # sixteen scalar kernel arguments + the mask words: 6 SGPRs go to VGPR lanes (v_writelane, no memory)
    ("bst_softmax_kernel", None): (0, 6),
}


def _base(name):
    """The function name inside an Itanium-mangled symbol (_ZN4bsmm14xflow32_kernelI...: length-prefixed identifiers)."""
    pos = name.find("_ZN")
    pos = pos + 3 if pos >= 0 else (2 if name.startswith("_Z") else None)
    if pos is None:
        return name
    last = name
    while pos < len(name):
        m = re.match(r"\d+", name[pos:])
        if not m:
            break
        n = int(m.group(0))
        last = name[pos + len(m.group(0)):pos + len(m.group(0)) + n]
        pos += len(m.group(0)) + n
        if last.endswith("_kernel"):
            break
    return last


@pytest.fixture(scope="module")
def meta():
    import __graft_entry__ as g
    g.build()
    return C.kernel_metadata(SO)


def test_every_gfx950_code_object_is_found(meta):
    assert len(meta) > 400                       # three translation units' worth of kernels
    objs = C.code_objects(SO)
    assert objs and all("gfx950" in t for t, _ in objs), [t for t, _ in objs]
    names = {_base(n) for n in meta}
    missing = [h for h in HOT if h not in names]
    assert not missing, missing                  # a renamed hot kernel must be renamed here too


def test_no_kernel_spills(meta):
    bad = []
    for name, k in sorted(meta.items()):
        v, s, scratch = k.get(".vgpr_spill_count", 0), k.get(".sgpr_spill_count", 0), k.get(".private_segment_fixed_size", 0)
        if not (v or s or scratch):
            continue
        base = _base(name)
        allowed = None
        for (kn, sub), lim in KNOWN.items():
            if kn == base and (sub is None or sub in name):
                allowed = lim
        if base in HOT or allowed is None or v > allowed[0] or s > allowed[1] or (scratch and not v):
            bad.append((name, v, s, scratch))
    assert not bad, bad


def test_full_cu_workgroups_fit_their_occupancy(meta):
    """A 1024-thread workgroup needs four waves per SIMD: at most 128 registers (architectural + accumulation) per lane."""
    for name, k in meta.items():
        if k.get(".max_flat_workgroup_size", 0) == 1024:
            total = ((k.get(".vgpr_count", 0) + 7) // 8) * 8 + k.get(".agpr_count", 0) if k.get(".agpr_count", 0) else k.get(".vgpr_count", 0)
            assert total <= 128, (name, k.get(".vgpr_count"), k.get(".agpr_count"))
