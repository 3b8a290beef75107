#!/bin/bash
# quick rocprofv3 kernel stats of the Tacotron pass (tuning aid; `gpurun -- bash scripts/tacotron_stats.sh [tag]`)
set -u
TAG=${1:-quick}
REPO=$PWD
OUT=$PWD/gpurun_out/tacotron_stats_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $REPO/scripts/tacotron_bench.py --steps 3 > $OUT/run.log 2>&1
tail -1 $OUT/run.log
python - "$OUT" <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("all kernels: %.2f ms over 4 passes -> %.2f ms per pass" % (tot / 1e6, tot / 4e6))
for r in rows[:12]:
    print("%-48s calls %4s  avg %9.1f us  total %7.2f ms  %5s %%" % (r["Name"][:48], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
