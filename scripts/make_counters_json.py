"""gpurun_out/prof (scripts/gpu_profile.sh) -> profiles/rNN_pmc.txt, rNN_kernel_trace_<workload>.md, rNN_counters.json, rNN_counters.md.

rNN_counters.json: {workload string of bench.py: {kernel label: {hbm_bytes, fetch_kib, write_kib, time_us, hbm_gbps, mfma_busy, waiting}}}
per LAUNCH.  hbm_bytes = 2 * FETCH_SIZE KiB (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE KiB; time_us =
GRBM_GUI_ACTIVE / 8 XCDs at 2.07 GHz (profiler attached: slower than the bench); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles * 1024
SIMDs).  bench.py quotes these next to its own timings when the workload string matches exactly."""
import json, os, shutil, sys
rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/prof"
WL = "bsmm fprop+bprop+updat %dx%d block_size=%d density=%d%% feature_axis=%d, minibatch %d per GPU, layout default_rng(1234)"
WORKLOADS = {"d10": WL % (4096, 4096, 32, 10, 1, 8192), "d20": WL % (4096, 4096, 32, 20, 1, 8192), "d50": WL % (4096, 4096, 32, 50, 1, 8192),
             "cfg2": WL % (4096, 4096, 16, 10, 0, 8192), "cfg3": WL % (8192, 8192, 32, 5, 1, 4096)}
BLOCKS = {"d10": (1667, 32, 8192), "d20": (3279, 32, 8192), "d50": (8210, 32, 8192), "cfg2": (6511, 16, 8192), "cfg3": (3220, 32, 4096)}
# kernel-name fragment -> label (the labels bench.py's roofline uses, plus the parts of the updat pass)
NAMES = (("xflow32_kernel<bsmm::DTbf16, false", "bsmm_xprop(bprop)"), ("xflow32_kernel<bsmm::DTbf16, true", "bsmm_xprop(fprop)"),
         ("updat32_a1_v2", "bsmm_updat_kernel"), ("updat2_reduce", "bsmm_updat_reduce"), ("updat16_rows_finalize", "bsmm_updat_reduce"), ("updat16_rows", "bsmm_updat_kernel"), ("updat16_win", "bsmm_updat_kernel"), ("updat_finalize", "bsmm_updat_reduce"),
         ("xcol32_v2_kernel<bsmm::DTbf16, false", "bsmm_xprop(bprop)"), ("xcol32_v2_kernel<bsmm::DTbf16, true", "bsmm_xprop(fprop)"),
         ("xcol16_list_kernel<bsmm::DTbf16, 0, true", "bsmm_xprop(fprop)"), ("xcol16_list_kernel<bsmm::DTbf16, 0, false", "bsmm_xprop(bprop)"),
         ("xcol16_list_kernel<bsmm::DTbf16, 1, true", "bsmm_xprop(fprop)"), ("xcol16_list_kernel<bsmm::DTbf16, 1, false", "bsmm_xprop(bprop)"),
         ("xcol16_v2_kernel", "bsmm_xprop(bprop)"), ("transpose_blocks", "bsmm_transpose_blocks"))
txt = open(os.path.join(src, "pmc.txt")).read()
shutil.copy(os.path.join(src, "pmc.txt"), "profiles/%s_pmc.txt" % rnd)
for k in WORKLOADS:
    f = os.path.join(src, "kernel_trace_%s.md" % k)
    if os.path.exists(f):
        shutil.copy(f, "profiles/%s_kernel_trace_%s.md" % (rnd, k))
if os.path.exists(os.path.join(src, "bench_line.json")):
    line = open(os.path.join(src, "bench_line.json")).read().strip().splitlines()[-1]
    open("profiles/%s_bench_line.json" % rnd, "w").write(line + "\n")
data, names = {}, {}
for sec in txt.split("## workload ")[1:]:
    head, body = sec.split("\n", 1)
    key = head.split()[0]
    cur = None
    for line in body.splitlines():
        if line.startswith("=="):
            cur = next((v for k, v in NAMES if k in line), None)
            if cur:
                names.setdefault(key, {}).setdefault(cur, set()).add(line[2:].split("(")[0].strip())
        elif cur:
            parts = line.split()
            if len(parts) == 2:
                data.setdefault(key, {}).setdefault(cur, {})[parts[0]] = float(parts[1])
out, rows = {}, []
for key, kernels in data.items():
    blocks, bs, N = BLOCKS[key]
    rec = {}
    for label, e in kernels.items():
        cyc = e.get("GRBM_GUI_ACTIVE", 0) / 8.0
        t_us = cyc / 2070.0
        hbm = int((2 * e.get("FETCH_SIZE", 0) + e.get("WRITE_SIZE", 0)) * 1024)
        rec[label] = {"fetch_kib": e.get("FETCH_SIZE"), "write_kib": e.get("WRITE_SIZE"), "hbm_bytes": hbm, "time_us": round(t_us, 1),
                      "hbm_gbps": round(hbm / t_us / 1e3, 0) if t_us else None,
                      "mfma_busy": round(e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024), 4) if cyc else None,
                      "waiting": round(e.get("SQ_WAIT_ANY", 0) / max(1.0, e.get("SQ_WAVE_CYCLES", 1)), 3),
                      "kernel_names": sorted(names.get(key, {}).get(label, []))}
    # the updat PASS bench.py times = the kernel + the pass that sums its partial sums (when there is one)
    k, r = rec.get("bsmm_updat_kernel"), rec.get("bsmm_updat_reduce")
    if k:
        tot_t = k["time_us"] + (r["time_us"] if r else 0)
        tot_b = k["hbm_bytes"] + (r["hbm_bytes"] if r else 0)
        rec["bsmm_updat"] = {"hbm_bytes": tot_b, "time_us": round(tot_t, 1), "hbm_gbps": round(tot_b / tot_t / 1e3, 0) if tot_t else None,
                             "mfma_busy": round(k["mfma_busy"] * k["time_us"] / tot_t, 4) if (tot_t and k["mfma_busy"] is not None) else None,
                             "waiting": k["waiting"], "kernel_names": k["kernel_names"] + (r["kernel_names"] if r else [])}
    out[WORKLOADS[key]] = rec
    for label in ("bsmm_xprop(fprop)", "bsmm_xprop(bprop)", "bsmm_updat_kernel", "bsmm_updat_reduce", "bsmm_updat"):
        e = rec.get(label)
        if e:
            tf = 2.0 * blocks * bs * bs * N / e["time_us"] / 1e6 if (e["time_us"] and "reduce" not in label) else 0
            rows.append("| %s | %s | %.1f | %.0f | %s | %s | %s | %s |" % (key, label, e["time_us"], e["hbm_bytes"] / 1e6, e["hbm_gbps"],
                        ("%.1f %%" % (100 * e["mfma_busy"])) if e.get("mfma_busy") is not None else "—", ("%.0f" % tf) if tf else "—",
                        ("%.0f %%" % (100 * e["waiting"])) if e.get("waiting") is not None else "—"))
sys.path.insert(0, os.getcwd())
import subprocess
import bench as _bench
out["_stamp"] = {"csrc_digest": _bench.kernel_sources_digest(),     # bench.py quotes these numbers only while the kernel sources are the ones profiled
                 "git_head_when_made": subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip(),
                 "note": "digest = sha256 over blocksparse_amd/csrc/*.{h,hip}; the GPU box that took the counters ran exactly this tree (gpurun snapshot)"}
out["_comment"] = __doc__
json.dump(out, open("profiles/%s_counters.json" % rnd, "w"), indent=1)
md = ["# Counters per workload (from `profiles/%s_pmc.txt`; MI355X, rocprofv3 --pmc, separate passes)" % rnd, "",
      "Per launch.  time = GRBM_GUI_ACTIVE / 8 XCDs at 2.07 GHz (profiler attached, 8 steps per run at cold clocks: 10-15 % slower than the bench line);",
      "HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB; gfx950 correction for the read side); MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs);",
      "effective TFLOP/s = 2 x blocks x bs² x N / time.  cfg2's xprop kernel is `xcol16_list_kernel` (TRANSW = true: fprop), its weight gradient `updat16_rows_kernel` + `updat16_rows_finalize_kernel` (round 5).", "",
      "| workload | kernel | time µs | HBM MB | HBM GB/s | MFMA busy | eff. TFLOP/s | waves waiting |", "|---|---|---|---|---|---|---|---|"] + rows
open("profiles/%s_counters.md" % rnd, "w").write("\n".join(md) + "\n")
print("\n".join(rows))
