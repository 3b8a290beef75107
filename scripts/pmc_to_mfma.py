#!/usr/bin/env python
"""Condense the rocprofv3 passes of scripts/profile_secondaries.sh into gpurun_out/prof_<tag>/summary_mfma_<tag>.txt: per kernel the
launch count and average duration (kernel-trace stats run) and, from the counter runs, the matrix-core occupancy.

Units (MI355X_MICROARCH.md, "Per-instruction cycle constants"): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles,
SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs.  v_mfma_f32_32x32x2_f32 keeps a SIMD's matrix pipe busy 64 cycles
for 4096 FLOP and v_mfma_f32_16x16x4_f32 32 cycles for 2048: 64 FLOP per busy cycle either way, so
    executed MFMA FLOP of a launch = 64 x SQ_VALU_MFMA_BUSY_CYCLES        (f32-input MFMA only: every MFMA in this repo)
    MFMA TFLOP/s = that / launch duration;   fraction of the 157.3 TFLOP/s f32 matrix peak beside it."""
import csv, glob, os, re, sys
out, tag = sys.argv[1], sys.argv[2]
lines = []


def rows(pattern):
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r


def short(name):
    m = re.search(r"(tc_\w+|tr_\w+|wn_\w+|Cijk_\w{0,40}|twv_\w+)", name)
    return m.group(1) if m else name[:48]


for work, title in (("tacotron", "configs[2] Tacotron text->mel, scripts/tacotron_bench.py --steps 3 (4 passes of B=32 x 200 decoder steps)"),
                    ("train", "configs[3] training step, scripts/train_bench.py --steps 3 --warmup 1 (4 steps of B=64 x 7800)")):
    stats = sorted(rows("stats_%s/**/*kernel_stats.csv" % work), key=lambda r: -float(r["TotalDurationNs"]))
    if not stats:
        lines.append("%s: no stats" % work)
        continue
    lines.append(title)
    lines.append("  rocprofv3 --kernel-trace --stats:")
    tot = sum(float(r["TotalDurationNs"]) for r in stats)
    for r in stats[:14]:
        lines.append("    %-44s calls %5s  avg %10.1f us  total %9.2f ms  %5.1f %%" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                                  float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
    lines.append("    (all kernels: %.2f ms)" % (tot / 1e6))
    # counters: sum per kernel over the launches of the run; durations of the SAME (profiled) run from its kernel trace
    agg = {}
    for r in rows("pmc_mfma_%s/**/*counter_collection.csv" % work):
        k = short(r["Kernel_Name"])
        a = agg.setdefault(k, {"_n": set(), "_dur": {}})
        a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        a["_n"].add(r["Dispatch_Id"])
        if "Start_Timestamp" in r and "End_Timestamp" in r:
            a["_dur"][r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    mops = {}
    for r in rows("pmc_mops_%s/**/*counter_collection.csv" % work):
        k = short(r["Kernel_Name"])
        mops.setdefault(k, {})
        mops[k][r["Counter_Name"]] = mops[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    avg = {short(r["Name"]): float(r["AverageNs"]) for r in stats}
    lines.append("  counter run (--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE), sums over the launches:")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)):
        busy = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if busy <= 0:
            continue
        n = len(a["_n"])
        dur_ns = sum(a["_dur"].values()) if a["_dur"] else avg.get(k, 0.0) * n
        src = "this run's trace" if a["_dur"] else "stats run avg x launches"
        wave = a.get("SQ_WAVE_CYCLES", 0.0)
        tf = 64.0 * busy / dur_ns / 1e3 if dur_ns else float("nan")
        lines.append("    %-28s launches %4d  MFMA_BUSY %.4e cyc  SQ_BUSY %.4e  WAVE %.4e qc  GUI_ACTIVE %.4e" % (k, n, busy, a.get("SQ_BUSY_CYCLES", 0.0), wave, a.get("GRBM_GUI_ACTIVE", 0.0)))
        lines.append("    %-28s   MFMA busy / (4 x SQ_WAVE_CYCLES) = %5.1f %%   WAIT_ANY/WAVE %.2f  WAIT_INST/WAVE %.2f  ACTIVE_VALU/WAVE %.2f" % (
            "", 100 * busy / (4 * wave) if wave else float("nan"), a.get("SQ_WAIT_ANY", 0) / wave if wave else 0, a.get("SQ_WAIT_INST_ANY", 0) / wave if wave else 0,
            a.get("SQ_ACTIVE_INST_VALU", 0) / wave if wave else 0))
        lines.append("    %-28s   executed MFMA FLOP = 64 x busy = %.4e over %.3f ms (%s) -> %.1f TFLOP/s = %.1f %% of 157.3" % ("", 64 * busy, dur_ns / 1e6, src, tf, 100 * tf / 157.3))
        if k in mops:
            lines.append("    %-28s   %s" % ("", "  ".join("%s %.4e" % kv for kv in sorted(mops[k].items()))))
    lines.append("")
txt = "\n".join(lines)
print(txt)
open(os.path.join(out, "summary_mfma_%s.txt" % tag), "w").write(txt + "\n")
