#!/usr/bin/env python
"""HBM bytes of one configs[3] training step from the FETCH_SIZE / WRITE_SIZE passes of scripts/train_traffic.sh ->
gpurun_out/train_traffic_<tag>/summary_train_traffic_<tag>.txt and profiles/traffic.json["train:" + _lib.train_hash()].
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md, HBM section).  The profiled command runs 2 steps.
The step's total counts the step's own kernels (tr_* and the rocBLAS Cijk_* products): the one-time clear of a new workspace and
torch's allocator fills belong to no step."""
import csv, glob, json, os, sys, collections
out, tag = sys.argv[1], sys.argv[2]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in sorted(glob.glob(out + "/pmc_%s/**/*counter_collection.csv" % c, recursive=True), key=os.path.getmtime)[-1:]:   # (the newest run)
        for r in csv.DictReader(open(f)):
            vals[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in sorted(glob.glob(out + "/pmc_%s/**/*kernel_trace.csv" % c, recursive=True), key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"][:48]].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
rows = []
for k, v in vals.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        n = len(v["FETCH_SIZE"])
        fe = 2 * 1024 * sum(v["FETCH_SIZE"]) / n; wr = 1024 * sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
        us = sum(dur[k]) / max(1, len(dur[k]))
        rows.append((us * n, k, n, us, fe, wr))
lines = ["HBM traffic of the training step's kernels (configs[3], B = 64 x 7800; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes with --kernel-trace "
         "only, KiB; FETCH_SIZE x 2: gfx950 correction; the profiled command runs 2 steps: per launch, next to the kernel's average duration in the counter pass)"]
for tot, k, n, us, fe, wr in sorted(rows, reverse=True)[:24]:
    lines.append("%-48s launches %4d  avg %8.1f us (under counters)  read %8.1f MB  written %8.1f MB  -> %5.2f TB/s" % (k, n, us, fe / 1e6, wr / 1e6, (fe + wr) / us / 1e6))
steps = 2.0
own = [r for r in rows if r[1].startswith(("tr_", "void tr_", "Cijk_"))]
fetch = sum(fe * n for _, _, n, _, fe, _ in own) / steps
write = sum(wr * n for _, _, n, _, _, wr in own) / steps
busy = sum(tot for tot, *_ in own) / steps
lines.append("=> one step: %.2f GB read + %.2f GB written; kernel time under counters %.2f ms -> %.2f TB/s averaged over the step's kernels"
             % (fetch / 1e9, write / 1e9, busy / 1e3, (fetch + write) / busy / 1e6))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import twvk_amd
h = "train:" + twvk_amd._lib.train_hash()
p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
t = json.load(open(p)) if os.path.exists(p) else {}
t.setdefault(h, {})["B64_T7800"] = {"step_fetch_bytes": fetch, "step_write_bytes": write, "profile": "profiles/%s_train_traffic_v2.txt" % tag}
json.dump(t, open(p, "w"), indent=1, sort_keys=True)
lines.append("(profiles/traffic.json updated under %s)" % h)
open(os.path.join(out, "summary_train_traffic_%s.txt" % tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
