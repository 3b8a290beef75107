#!/bin/bash
# WRITE_SIZE of the many-streams kernel at B = 64 for two builds of the library (tuning aid: lc ring length)
REPO=$PWD; OUT=$PWD/gpurun_out/ring_pmc; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for v in ring16 ring8; do
  TWV_AMD_LIB=$REPO/scripts/libtwv_$v.so.bin rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/$v -- python $REPO/scripts/many_bench.py --batch 64 --seconds 0.5 --steps 1 --warmup 1 > $OUT/$v.log 2>&1
  python - "$OUT/$v" $v <<'PY'
import csv, glob, sys
tot = 0.0; n = 0
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wn_xcd_many_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE":
            tot += float(r["Counter_Value"]); n += 1
print(sys.argv[2], "launches", n, "WRITE_SIZE KiB per launch", tot / max(n, 1), "-> bytes per step (12000 steps):", tot / max(n, 1) * 1024 / 12000)
PY
done
