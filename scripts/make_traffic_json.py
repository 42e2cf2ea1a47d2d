"""profiles/rNN_pmc.txt (scripts/gpu_profile.sh) -> profiles/rNN_traffic.json: HBM bytes per launch of the bench-path kernels,
per density.  read = 2 * FETCH_SIZE KiB (gfx950 correction, MI355X_MICROARCH.md HBM section), write = WRITE_SIZE KiB."""
import json, sys
rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
txt = open("profiles/%s_pmc.txt" % rnd).read().split("## density ")[1:]
NAMES = (("updat32_a1_v2", "bsmm_updat_kernel"), ("updat2_reduce", "bsmm_updat_reduce"), ("updat_finalize", "bsmm_updat_finalize"),
         ("xcol32_v2_kernel<bsmm::DTbf16, false", "bsmm_xprop(bprop)"), ("xcol32_v2_kernel<bsmm::DTbf16, true", "bsmm_xprop(fprop)"),
         ("xcol32_a1_v2_kernel<bsmm::DTbf16, false", "bsmm_xprop(bprop)"), ("xcol32_a1_v2_kernel<bsmm::DTbf16, true", "bsmm_xprop(fprop)"),
         ("xcol32_a1_kernel", "bsmm_xprop_round1"), ("transpose_blocks", "bsmm_transpose_blocks"))
data = {}
for sec in txt:
    head, body = sec.split("\n", 1)
    key = "d%d" % round(float(head.split()[0]) * 100)
    cur = None
    for line in body.splitlines():
        if line.startswith("=="):
            cur = next((v for k, v in NAMES if k in line), None)
        elif cur and ("FETCH_SIZE" in line or "WRITE_SIZE" in line):
            k, v = line.split()
            data.setdefault(key, {}).setdefault(cur, {})["fetch_kib" if k == "FETCH_SIZE" else "write_kib"] = float(v)
wl = "bsmm fprop+bprop+updat 4096x4096 block_size=32 density=%d%% feature_axis=1, minibatch 8192 per GPU, layout default_rng(1234)"
for v in data.values():
    for e in v.values():
        e["hbm_bytes"] = int((2 * e.get("fetch_kib", 0) + e.get("write_kib", 0)) * 1024)
    # the updat PASS bench.py times = the streaming kernel + its reduce pass (no reduce pass when the kernel stores directly)
    k, r = v.get("bsmm_updat_kernel"), v.get("bsmm_updat_reduce", {})
    if k:
        v["bsmm_updat"] = {"fetch_kib": k.get("fetch_kib", 0) + r.get("fetch_kib", 0), "write_kib": k.get("write_kib", 0) + r.get("write_kib", 0),
                           "hbm_bytes": k["hbm_bytes"] + r.get("hbm_bytes", 0)}
out = {"_comment": "HBM bytes per launch from rocprofv3 PMC passes (profiles/%s_pmc.txt, scripts/gpu_profile.sh): read = 2 * FETCH_SIZE KiB "
                   "(gfx950 correction, MI355X_MICROARCH.md HBM section), write = WRITE_SIZE KiB.  One FETCH_SIZE pass and one WRITE_SIZE "
                   "pass per density.  bsmm_updat = the updat pass bench.py times: the streaming kernel (bsmm_updat_kernel) + the pass that sums its partial sums (bsmm_updat_reduce)." % rnd,
       "workload": wl % 20}
out.update(data.get("d20", {}))
out["densities"] = {k: dict(workload=wl % int(k[1:]), **v) for k, v in data.items()}
json.dump(out, open("profiles/%s_traffic.json" % rnd, "w"), indent=1)
print({k: {n: e["hbm_bytes"] for n, e in v.items()} for k, v in data.items()})
