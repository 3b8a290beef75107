#!/bin/bash
# HBM traffic of the training step's kernels (tuning aid; `gpurun -- bash scripts/train_traffic.sh [tag]`): FETCH_SIZE and WRITE_SIZE in
# separate passes (KiB; FETCH_SIZE x 2 on gfx950, MI355X_MICROARCH.md), printed per launch next to the kernel's average duration.
set -u
TAG=${1:-quick}
REPO=$PWD
OUT=$PWD/gpurun_out/train_traffic_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -- python $REPO/scripts/train_bench.py --steps 1 --warmup 1 > $OUT/pmc_$C.log 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(out + "/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            vals[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(out + "/pmc_%s/**/*kernel_trace.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"][:48]].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
rows = []
for k, v in vals.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        n = len(v["FETCH_SIZE"])
        fe = 2 * 1024 * sum(v["FETCH_SIZE"]) / n; wr = 1024 * sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
        us = sum(dur[k]) / max(1, len(dur[k]))
        rows.append((us * n, k, n, us, fe, wr))
for tot, k, n, us, fe, wr in sorted(rows, reverse=True)[:16]:
    print("%-48s launches %4d  avg %8.1f us (under counters)  read %8.1f MB  written %8.1f MB  -> %5.2f TB/s" % (k, n, us, fe / 1e6, wr / 1e6, (fe + wr) / us / 1e6))
PY
