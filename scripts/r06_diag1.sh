#!/bin/bash
# round 6, first GPU call: where the x decoder and the training step stand before this round's work
set -u
OUT=$PWD/gpurun_out/r06_diag1; mkdir -p $OUT; REPO=$PWD
python scripts/tacotron_xdec_profile.py > $OUT/xdec_profile_before.txt 2>&1
python scripts/tacotron_bench.py --steps 5 > $OUT/taco_bench_default.json 2>&1
python scripts/tacotron_bench.py --steps 5 --decoder-groups 32 > $OUT/taco_bench_x.json 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/train_trace -- python $REPO/scripts/train_bench.py --steps 2 --warmup 1 > $OUT/train_trace.log 2>&1
cd $REPO
python - <<'P' > $OUT/train_trace_order.txt 2>&1
import csv, glob
f = glob.glob("gpurun_out/r06_diag1/train_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    n = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "Fill" in n or "fill" in n or d > 150:
        print("%10.1f us  %8.1f us  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, d, n[:100]))
P
rm -rf $OUT/train_trace
