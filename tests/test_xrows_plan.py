"""Host-side checks of the 'BSX5' plans of the row-split xprop kernel (csrc/bsmm_plan.h build_xrows_plan, csrc/bsmm_xrows.h).  The records
are the whole synchronisation contract of that kernel (one barrier per step, counted vmcnt waits, DMA duties), so they are validated here
WITHOUT a GPU by replaying them against a model of the LDS ring:
  * every lut entry of an output column appears exactly once in the block masks, in ascending input-block order per column;
  * a step's blocks sit in consecutive weight slots, in mask-bit order, and hold the right weight block when the step runs;
  * the step's slab slot holds the step's pair when the step runs;
  * a duty never targets a slot the CURRENT step reads (every other reader is covered by the content checks: the model updates a slot's
    content when the request is issued), the prologue never touches the staging slab (slot X5_D - 1);
  * the vmcnt of every (step, wave) retires the loads the NEXT step reads (everything lands one barrier early: the kernel requests a
    step's first fragments while it finishes the one before), step 0's also its own: replaying the pair's issue order (loads complete in order), none
    of the loads that may still be in flight targets a slot the step reads, and no slot is requested again while an older request for it
    may be in flight."""
import numpy as np
import pytest

import _parity as P
from blocksparse_amd import _lib as lib
from blocksparse_amd import lut as L
from blocksparse_amd.matmul import _host_plan

MAGIC = 0x42535835
REC = 64


def _plan(layout, which):
    t = L.build_tables(layout, z_order=True, segmented=False)
    side = t[which]
    n_out = t["KB"] if which == "fprop" else t["CB"]
    words = _host_plan(side["lut"], side["segments"], t["blocks"], n_out, 32, lib.BF16, 1, lib.PLAN_XCOL_ROWS)
    return t, side, n_out, words


def _columns(side):
    lut = np.asarray(side["lut"])
    cols = {}
    for s in range(side["segments"]):
        off, cnt, ob, _ = lut[4 * s:4 * s + 4]
        ent = lut[2 * off:2 * (off + cnt)].reshape(-1, 2)
        cols.setdefault(int(ob), []).extend((int(c), int(w)) for c, w in ent)
    return cols


LAYOUTS = [("random 40x24", P.random_layout(40, 24, 0.3, seed=2)), ("dense 12x20", np.ones((12, 20), dtype=np.int32)),
           ("dense 64x33", np.ones((64, 33), dtype=np.int32)), ("BA 64", P.ba_layout(64, 5, seed=1)),
           ("sparse 300x16", P.random_layout(300, 16, 0.05, seed=6)), ("single", np.ones((1, 1), dtype=np.int32)),
           ("groups without blocks", np.eye(15, 40, dtype=np.int32)), ("half 33x47", P.random_layout(33, 47, 0.5, seed=3)),
           ("bench 20 %", P.random_layout(128, 128, 0.2, seed=1234)), ("bench 50 %", P.random_layout(128, 128, 0.5, seed=5))]


@pytest.mark.parametrize("name,layout", LAYOUTS)
@pytest.mark.parametrize("which", ["fprop", "bprop"])
def test_rows_plan_replay(name, layout, which):
    t, side, n_out, p = _plan(layout, which)
    assert p is not None and p[0] == MAGIC and p[2] == 16
    D, NW, CAP, PRO = p[7] & 0xff, (p[7] >> 8) & 0xff, (p[7] >> 16) & 0xff, (p[7] >> 24) & 0xff
    assert PRO == D - 1 and NW >= 3 * CAP + 8
    cols = _columns(side)
    ngroups, off_groups, off_recs = p[3], p[5], p[6]
    assert off_recs % 4 == 0 and len(p) == off_recs + (p[4] + 1) * REC       # (+ the padding record)
    seen_groups = set()
    total_blocks = 0
    for g in range(ngroups):
        rec_off, nsteps, ob0, nob, nblk = p[off_groups + 8 * g:off_groups + 8 * g + 5]
        assert ob0 % 16 == 0 and ob0 not in seen_groups and nob == min(16, n_out - ob0)
        seen_groups.add(ob0)
        recs = p[off_recs + rec_off * REC:off_recs + (rec_off + PRO + nsteps) * REC].reshape(PRO + nsteps, REC)
        slab = [None] * D                   # content: pair index
        wslot = [None] * NW                 # content: weight block id
        issued = [[] for _ in range(4)]     # per wave: the loads in issue order, as ("x", slab slot) / ("w", weight slot)
        done = [0] * 4                      # ... and how many of them its waits have retired
        col_seq = {ob0 + c: [] for c in range(16)}
        last_pair = -1

        def reads_of(rc):
            xs = int(rc[1]) // 16384
            m, s0 = int(rc[2]) & 0xffffffff, int(rc[3])
            return xs, m, s0, set(range(s0, s0 + bin(m).count("1")))
        for rr in range(PRO + nsteps):
            rc = recs[rr]
            step = rr - PRO
            reads_x, reads_w = None, set()
            if step >= 0:
                pair = int(rc[0])
                xs, m, s, reads_w = reads_of(rc)
                assert pair >= last_pair and 0 <= xs < D and int(rc[1]) % 16384 == 0
                last_pair = pair
                # ---- what the step reads ----
                assert slab[xs] == pair, (g, step, "slab slot holds another pair")
                reads_x = xs
                assert 0 < len(reads_w) <= CAP and s + len(reads_w) <= NW
                for bit in range(32):
                    if not (m >> bit) & 1:
                        continue
                    ob, c = ob0 + (bit >> 1), 2 * pair + (bit & 1)
                    assert ob < ob0 + nob
                    want = dict(cols.get(ob, [])).get(c)
                    assert want is not None and wslot[s] == want, (g, step, bit, "weight slot holds another block")
                    col_seq[ob].append((c, want))
                    s += 1
                total_blocks += len(reads_w)
                # ---- the waits (in front of this step's barrier): what the NEXT step reads (step 0: and this one) has landed; loads complete
                # in order, so what an earlier wait retired stays retired ----
                nxt_x, nxt_w = None, set()
                if step + 1 < nsteps:
                    nrc = recs[rr + 1]
                    assert int(rc[8]) == int(nrc[1]) and int(rc[9]) == int(nrc[2]) and int(rc[10]) == int(nrc[3])
                    nxt_x, _, _, nxt_w = reads_of(nrc)
                else:
                    assert int(rc[9]) == 0
                for q in range(4):
                    n = (int(rc[4]) >> (8 * q)) & 0xff
                    assert n <= 63
                    done[q] = max(done[q], len(issued[q]) - n)
                    assert len(issued[q]) - done[q] <= 63, "more loads in flight than the counter holds"
                    for kind, slot in issued[q][done[q]:]:
                        assert not (kind == "x" and slot == nxt_x) and not (kind == "w" and slot in nxt_w), (g, step, q, n, "wait too weak (next step)")
                        assert not (kind == "x" and slot == reads_x) and not (kind == "w" and slot in reads_w), (g, step, q, n, "read before it landed")
            else:
                assert all(int(v) == 0 for v in rc[:5])
            # ---- duties of the record (issued behind the step's barrier) ----
            xp = int(rc[5])
            if xp >= 0:
                xs = int(rc[6]) // 16384
                assert int(rc[6]) % 16384 == 0 and 0 <= xs < D and xs != reads_x and (step >= 0 or xs < D - 1), (g, rr, "slab duty hits a slot in use")
                slab[xs] = xp
                for q in range(4):
                    assert ("x", xs) not in issued[q][done[q]:], (g, rr, "slab slot requested again while an older request may be in flight")
                    issued[q] += [("x", xs)] * 4
            ents = rc[16:48].reshape(4, 4, 2)
            for e in range(16):
                off, wo = int(ents[e & 3][e >> 2][0]), int(ents[e & 3][e >> 2][1]) & 0xffffffff
                if off < 0:
                    assert all(int(ents[k & 3][k >> 2][0]) < 0 for k in range(e, 16))      # the list is packed
                    break
                assert (off - D * 16384) % 2048 == 0 and wo % 2048 == 0
                w, s = wo >> 11, (off - D * 16384) // 2048
                assert 0 <= s < NW and s not in reads_w, (g, rr, "weight duty hits a slot in use")
                wslot[s] = w
                assert all(("w", s) not in issued[k][done[k]:] for k in range(4)), (g, rr, "weight slot requested again while an older request may be in flight")
                issued[e & 3] += [("w", s)] * 2
        for ob, seq in col_seq.items():
            assert seq == sorted(cols.get(ob, [])), (g, ob, "column blocks missing / out of order")
        assert sum(len(s) for s in col_seq.values()) == nblk
    assert total_blocks == t["blocks"]
    assert all(int(v) == 0 for v in p[off_recs + p[4] * REC:])


def test_rows_plan_is_refused_for_other_axis_and_dtype():
    t = L.build_tables(P.random_layout(8, 8, 0.5, seed=1), z_order=True, segmented=False)
    side = t["fprop"]
    w0 = _host_plan(side["lut"], side["segments"], t["blocks"], t["KB"], 32, lib.BF16, 0, lib.PLAN_XCOL_ROWS)
    assert w0 is None or w0[0] != MAGIC          # feature axis 0: the option is ignored (the staged plan)
    w1 = _host_plan(side["lut"], side["segments"], t["blocks"], t["KB"], 32, lib.F32, 1, lib.PLAN_XCOL_ROWS)
    assert w1 is None or w1[0] != MAGIC


def test_rows_kernel_keeps_its_reserved_registers(tmp_path):
    """csrc/bsmm_xrows.h keeps a step's fragments in v[216:255] ACROSS its asm statements (requested by one step's exit, multiplied by the
    next).  Nothing reserves those registers from the compiler but its own appetite (the accumulators live in AGPRs, its code needs < 100
    VGPRs): an earlier form of the kernel, with the accumulators in VGPRs, had an experiment build move them through the reserved range.
    This audit compiles the kernel alone (device only, seconds) and requires that every instruction touching v216+ is one of the asm
    statements' own (fragment reads, MFMA operands), no scratch, no spill."""
    import os, re, shutil, subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(hipcc) and os.path.exists(objdump)):
        pytest.skip("no hipcc / llvm-objdump here")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "xrows_audit.hip"
    src.write_text('#include "bsmm_xrows.h"\nusing namespace bsmm;\n' + "".join(
        "template __global__ void bsmm::xrows32_kernel<%s, %s>(const uint16_t*, const uint16_t*, uint16_t*, const int32_t*, XMap, int, int, int);\n" % (dt, tw)
        for dt in ("DTbf16", "DTf16") for tw in ("false", "true")))
    obj = tmp_path / "xrows_audit.o"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "blocksparse_amd", "csrc"),
                        "--cuda-device-only", "--no-gpu-bundle-output", "-c", "-o", str(obj), str(src), "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "ScratchSize [bytes/lane]: 0" in r.stderr and not re.search(r"VGPRs Spill: [1-9]", r.stderr) and not re.search(r"ScratchSize \[bytes/lane\]: [1-9]", r.stderr), r.stderr[-1500:]
    dis = subprocess.run([objdump, "-d", str(obj)], capture_output=True, text=True).stdout
    own = ("ds_read_b128", "ds_read_b64_tr_b16", "v_mfma_f32_32x32x16_bf16", "v_mfma_f32_32x32x16_f16")
    bad = []
    for ln in dis.splitlines():
        body = ln.split("//")[0]
        regs = [int(x) for x in re.findall(r"\bv\[?(\d+)", body)]
        hi = [int(b) for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", body)]
        if any(x >= 216 for x in regs + hi) and (body.split() or [""])[0] not in own:
            bad.append(ln.strip()[:100])
    assert not bad, bad[:5]
    assert dis.count("s_ff1_i32_b64") >= 4 * 97


def test_rows_option_with_bsize64_nests_the_usual_plans():
    """BSMM_PLAN_XCOL_ROWS with bsize 64: the composite 'BS64' plan nests a plan the composite call can run (flow / staged), never 'BSX5',
    and attaches."""
    import ctypes
    lay = P.random_layout(6, 10, 0.4, seed=3)
    t = L.build_tables(lay, z_order=True, segmented=False)
    f = t["fprop"]
    plan = _host_plan(f["lut"], f["segments"], t["blocks"], t["KB"], 64, lib.BF16, 1, lib.PLAN_XCOL_ROWS)
    assert plan is not None and plan[0] == 0x42533634
    nested = plan[plan[5]:]
    assert int(nested[0]) != MAGIC
    a = lib.BsmmArgs()
    dev = np.zeros(plan.size + 4, dtype=np.int32)                 # (a host buffer stands in for the device copy: attach only records the pointer)
    addr = (dev.ctypes.data + 15) & ~15
    ip = ctypes.POINTER(ctypes.c_int32)
    assert lib.load().bsmm_plan_attach(ctypes.byref(a), plan.ctypes.data_as(ip), plan.size, ctypes.c_void_p(addr)) == 0
