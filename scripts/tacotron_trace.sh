#!/bin/bash
# per-launch durations of ONE Tacotron pass in launch order (tuning aid): rocprofv3 kernel trace of scripts/tacotron_bench.py
set -u
TAG=${1:-quick}
REPO=$PWD
OUT=$PWD/gpurun_out/tacotron_trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $REPO/scripts/tacotron_bench.py --steps 1 ${2:-} ${3:-} > $OUT/run.log 2>&1
tail -1 $OUT/run.log
python - "$OUT" <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the last pass: from the last tc_embed_kernel on
last = max(i for i, n in enumerate(names) if n.startswith("tc_embed_kernel"))
tot = {}
print("launch order of the last pass (us):")
for r in rows[last:]:
    n = r["Kernel_Name"].split("(")[0]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot[n] = tot.get(n, 0.0) + d
    print("  %-28s grid %6s x %-4s wg %4s  %9.1f" % (n[:28], r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"], d))
for n, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print("%-32s %9.1f us" % (n, v))
PY
