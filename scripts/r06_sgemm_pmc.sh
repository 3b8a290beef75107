#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r06_sgemm_pmc; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p1 -- $REPO/scripts/ubench/sgemm3.exe 300736 512 960 512 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/p2 -- $REPO/scripts/ubench/sgemm3.exe 300736 512 960 512 > $OUT/p2.log 2>&1
cd $REPO
python - <<'P'
import csv, glob, collections
for d in ("p1", "p2"):
    fs = glob.glob("gpurun_out/r06_sgemm_pmc/%s/**/*counter_collection.csv" % d, recursive=True)
    if not fs: print(d, "no counter file"); continue
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(fs[0])):
        if "sgemm" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in acc: print(d, k, acc[k] / max(n[k], 1))
P
tail -2 $OUT/p1.log
rm -rf $OUT/p1 $OUT/p2
