#!/usr/bin/env python
"""Condense the rocprofv3 passes of scripts/profile_generation.sh: per-kernel stats, FETCH_SIZE / WRITE_SIZE / SQ counters of the
generation kernel -> gpurun_out/prof_<tag>/summary_<tag>.txt, and HBM bytes per generation step -> profiles/traffic.json (keyed by
the hash of the generation kernels' sources: bench.py reports `roofline.traffic` only for the code it was measured on).

FETCH_SIZE / WRITE_SIZE are in KiB; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 counts 64 B per 128-B request of a
wide coalesced read -> doubled; WRITE_SIZE is taken as reported (uncalibrated)."""
import csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out, tag = sys.argv[1], sys.argv[2]
lines = []


def rows(pattern):
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r


stats = sorted(rows("stats/**/*kernel_stats.csv"), key=lambda r: -float(r["TotalDurationNs"]))
lines.append("rocprofv3 --kernel-trace --stats, bench.py --seconds 1.0 --steps 3 --warmup 1 (24 000 generation steps per launch)")
for r in stats[:8]:
    lines.append("  %-90s calls %3s  avg %12.1f us  %6s %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
gen = [r for r in stats if "generate_kernel" in r["Name"]]
if gen:
    lines.append("  => %s: %.3f us per generation step" % (re.search(r"wn_\w+", gen[0]["Name"]).group(0), float(gen[0]["AverageNs"]) / 1e3 / 24000))
steps = 12000
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ"):
    for r in rows("pmc_%s/**/*counter_collection.csv" % c):
        if "generate_kernel" in r["Kernel_Name"]:
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            vals["_name"] = re.search(r"wn_\w+", r["Kernel_Name"]).group(0)
            vals["_vgpr"] = r.get("VGPR_Count"); vals["_scratch"] = r.get("Scratch_Size"); vals["_lds"] = r.get("LDS_Block_Size")
lines.append("")
lines.append("PMC passes (separate runs, --kernel-trace only), bench.py --seconds 0.5 --steps 1 --warmup 0 (12 000 generation steps), kernel %s" % vals.get("_name"))
lines.append("  VGPRs %s, scratch %s B/lane, LDS %s B" % (vals.get("_vgpr"), vals.get("_scratch"), vals.get("_lds")))
for k, v in sorted(vals.items()):
    if not k.startswith("_"):
        lines.append("  %-22s %18.1f" % (k, sum(v) / len(v)))
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    fetch = 2.0 * 1024.0 * sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"]) / steps
    write = 1024.0 * sum(vals["WRITE_SIZE"]) / len(vals["WRITE_SIZE"]) / steps
    lines.append("  => HBM traffic per generation step (all 8 streams): %.0f B read (FETCH_SIZE KiB x 2, gfx950 correction) + %.0f B written" % (fetch, write))
    import twvk_amd
    h = twvk_amd._lib.generation_hash()
    p = os.path.join(ROOT, "profiles", "traffic.json")
    t = json.load(open(p)) if os.path.exists(p) else {}
    kname = "wn_xcd_generate_kernel" if "xcd" in vals["_name"] else "wn_generate_kernel"
    t.setdefault(h, {}).setdefault(kname, {})["B8_NL30"] = {"fetch_bytes_per_step": fetch, "write_bytes_per_step": write, "profile": "profiles/%s_rocprofv3_generation_summary.txt" % tag}
    json.dump(t, open(p, "w"), indent=1, sort_keys=True)
    lines.append("  (profiles/traffic.json updated for generation-kernel source hash %s)" % h)
# ---- the many-streams kernel at batch 64
mstats = sorted(rows("stats_many/**/*kernel_stats.csv"), key=lambda r: -float(r["TotalDurationNs"]))
mg = [r for r in mstats if "many_kernel" in r["Name"]]
if mg:
    lines.append("")
    lines.append("many-streams kernel, scripts/many_bench.py --batch 64 --seconds 1.0 --steps 3 --warmup 1 (24 000 generation steps per launch, 64 streams):")
    lines.append("  %-60s calls %3s  avg %12.1f us  => %.3f us per generation step, %.3f M samples/s" % (
        "wn_xcd_many_kernel", mg[0]["Calls"], float(mg[0]["AverageNs"]) / 1e3, float(mg[0]["AverageNs"]) / 1e3 / 24000, 64 * 24000 / (float(mg[0]["AverageNs"]) / 1e9) / 1e6))
mv = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in rows("pmc_many_%s/**/*counter_collection.csv" % c):
        if "many_kernel" in r["Kernel_Name"]:
            mv.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
if "FETCH_SIZE" in mv and "WRITE_SIZE" in mv:
    fetch = 2.0 * 1024.0 * sum(mv["FETCH_SIZE"]) / len(mv["FETCH_SIZE"]) / steps
    write = 1024.0 * sum(mv["WRITE_SIZE"]) / len(mv["WRITE_SIZE"]) / steps
    lines.append("  PMC (12 000-step launch): FETCH_SIZE %.1f KiB, WRITE_SIZE %.1f KiB => %.0f B read (x2 gfx950 correction) + %.0f B written per generation step (all 64 streams)" % (
        sum(mv["FETCH_SIZE"]) / len(mv["FETCH_SIZE"]), sum(mv["WRITE_SIZE"]) / len(mv["WRITE_SIZE"]), fetch, write))
    import twvk_amd
    h = twvk_amd._lib.generation_hash()
    p = os.path.join(ROOT, "profiles", "traffic.json")
    t = json.load(open(p)) if os.path.exists(p) else {}
    t.setdefault(h, {}).setdefault("wn_xcd_many_kernel", {})["B64_NL30"] = {"fetch_bytes_per_step": fetch, "write_bytes_per_step": write, "profile": "profiles/%s_rocprofv3_generation_summary.txt" % tag}
    json.dump(t, open(p, "w"), indent=1, sort_keys=True)
# ---- the one-hot mu-law-256 model (scripts/mulaw_bench.py: a 600-step warm-up launch + the 12 000-step launch, same kernel name)
qstats = sorted(rows("stats_mulaw/**/*kernel_stats.csv"), key=lambda r: -float(r["TotalDurationNs"]))
qg = [r for r in qstats if "generate_kernel" in r["Name"]]
if qg:
    lines.append("")
    lines.append("one-hot mu-law-256 model on the XCD kernel, scripts/mulaw_bench.py --batch 8 --steps 12000 (launches of 600 + 12 000 steps):")
    lines.append("  %-60s calls %3s  total %12.1f us  max %12.1f us => %.3f us per generation step (the 12 000-step launch)" % (
        re.search(r"wn_\w+", qg[0]["Name"]).group(0) + " (ONEHOT)", qg[0]["Calls"], float(qg[0]["TotalDurationNs"]) / 1e3, float(qg[0]["MaxNs"]) / 1e3,
        float(qg[0]["MaxNs"]) / 1e3 / 12000))
qv = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in rows("pmc_mulaw_%s/**/*counter_collection.csv" % c):
        if "generate_kernel" in r["Kernel_Name"]:
            qv.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
if "FETCH_SIZE" in qv and "WRITE_SIZE" in qv:
    qsteps = 12600.0
    fetch = 2.0 * 1024.0 * sum(qv["FETCH_SIZE"]) / qsteps
    write = 1024.0 * sum(qv["WRITE_SIZE"]) / qsteps
    lines.append("  PMC (both launches, 12 600 steps): FETCH_SIZE %.1f KiB, WRITE_SIZE %.1f KiB => %.0f B read (x2 gfx950 correction) + %.0f B written per generation step (all 8 streams)" % (
        sum(qv["FETCH_SIZE"]), sum(qv["WRITE_SIZE"]), fetch, write))
    import twvk_amd
    h = twvk_amd._lib.generation_hash()
    p = os.path.join(ROOT, "profiles", "traffic.json")
    t = json.load(open(p)) if os.path.exists(p) else {}
    t.setdefault(h, {}).setdefault("wn_xcd_generate_kernel_onehot", {})["B8_NL30"] = {"fetch_bytes_per_step": fetch, "write_bytes_per_step": write, "profile": "profiles/%s_rocprofv3_generation_summary.txt" % tag}
    json.dump(t, open(p, "w"), indent=1, sort_keys=True)
if "SQ_WAVE_CYCLES" in vals:
    wc = sum(vals["SQ_WAVE_CYCLES"]) / len(vals["SQ_WAVE_CYCLES"])
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
        if k in vals:
            lines.append("  %-22s / SQ_WAVE_CYCLES = %.3f" % (k, sum(vals[k]) / len(vals[k]) / wc))
txt = "\n".join(lines)
print(txt)
open(os.path.join(out, "summary_%s.txt" % tag), "w").write(txt + "\n")
