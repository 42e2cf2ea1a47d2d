"""profiles/rNN_pmc.txt -> profiles/rNN_counters_by_density.md: time, HBM bytes / rate, MFMA busy and effective TFLOP/s per kernel and density"""
import sys
rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
txt = open("profiles/%s_pmc.txt" % rnd).read().split("## density ")[1:]
NAMES = (("updat32_a1_v2", "updat kernel"), ("updat2_reduce", "updat reduce pass"), ("xcol32_v2_kernel<bsmm::DTbf16, false", "bprop"),
         ("xcol32_v2_kernel<bsmm::DTbf16, true", "fprop"))
data = {}
for sec in txt:
    head, body = sec.split("\n", 1)
    d = round(float(head.split()[0]) * 100)
    cur = None
    for line in body.splitlines():
        if line.startswith("=="):
            cur = next((v for k, v in NAMES if k in line), None)
        elif cur:
            parts = line.split()
            if len(parts) == 2:
                data.setdefault(d, {}).setdefault(cur, {})[parts[0]] = float(parts[1])
blocks = {10: 1667, 20: 3279, 50: 8210}
out = ["# Round-2 counters per density (from `profiles/%s_pmc.txt`; bf16, 4096², bs 32, feature_axis 1, N = 8192, MI355X)" % rnd, "",
       "Per launch.  time = GRBM_GUI_ACTIVE / 8 XCDs at 2.07 GHz; HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB; gfx950 correction for the read side);",
       "MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs); effective TFLOP/s = 2 x blocks x 32² x 8192 / time.  The counter runs are",
       "slower than the bench (profiler attached, ~8 steps per run at cold clocks): the bench line and `%s_kernel_trace.md` have the timings." % rnd, "",
       "| density | kernel | time µs | HBM MB | HBM GB/s | MFMA busy | eff. TFLOP/s | waves waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES) |", "|---|---|---|---|---|---|---|---|"]
for d in sorted(data):
    for k in ("fprop", "bprop", "updat kernel", "updat reduce pass"):
        e = data[d].get(k)
        if not e:
            continue
        cyc = e.get("GRBM_GUI_ACTIVE", 0) / 8.0
        t_us = cyc / 2070.0
        hbm = (2 * e.get("FETCH_SIZE", 0) + e.get("WRITE_SIZE", 0)) * 1024 / 1e6
        mf = e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024) if cyc else 0
        tf = 2.0 * blocks[d] * 1024 * 8192 / t_us / 1e6 if (t_us and "reduce" not in k) else 0
        wait = e.get("SQ_WAIT_ANY", 0) / max(1.0, e.get("SQ_WAVE_CYCLES", 1))
        out.append("| %d %% | %s | %.1f | %.0f | %.0f | %.1f %% | %s | %.0f %% |" % (d, k, t_us, hbm, hbm / t_us * 1e3 if t_us else 0, 100 * mf,
                                                                                     ("%.0f" % tf) if tf else "—", 100 * wait))
open("profiles/%s_counters_by_density.md" % rnd, "w").write("\n".join(out) + "\n")
print("\n".join(out[7:]))
