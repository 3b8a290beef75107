#!/bin/bash
# round 6: every matrix-core launch of one Tacotron configs[2] pass in launch order (kernel, grid, duration) -- rocprofv3 kernel trace
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/r06_gemm_trace; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python $REPO/scripts/tacotron_bench.py --steps 1 > $OUT/log.txt 2>&1
cd $REPO
python - <<'P'
import csv, glob
f = glob.glob("gpurun_out/r06_gemm_trace/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last pass = from the last tc_embed_kernel on
last = max(i for i, r in enumerate(rows) if "tc_embed_kernel" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
out = []
for r in rows[last:]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    out.append("%9.1f us  +%7.1f us  %-40s grid %s x %s x %s  wg %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, d, n[:40], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"]))
open("gpurun_out/r06_gemm_trace/launches.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
P
