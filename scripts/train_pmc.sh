#!/bin/bash
# SQ stall counters of the training step's kernels (tuning aid; `gpurun -- bash scripts/train_pmc.sh [tag] [kernel substring]`)
set -u
TAG=${1:-quick}; PAT=${2:-tr_layer}
REPO=$PWD
OUT=$PWD/gpurun_out/train_pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/p1 -- python $REPO/scripts/train_bench.py --steps 1 --warmup 1 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p2 -- python $REPO/scripts/train_bench.py --steps 1 --warmup 1 > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL --kernel-trace --output-format csv -d $OUT/p3 -- python $REPO/scripts/train_bench.py --steps 1 --warmup 1 > $OUT/p3.log 2>&1
python - "$OUT" "$PAT" <<'PY'
import csv, glob, sys, collections
out, pat = sys.argv[1], sys.argv[2]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            vals[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in vals.items():
    print(k)
    wc = sum(v["SQ_WAVE_CYCLES"]) / len(v["SQ_WAVE_CYCLES"]) if "SQ_WAVE_CYCLES" in v else 1.0
    for c in sorted(v):
        m = sum(v[c]) / len(v[c])
        print("    %-32s %14.4g   / WAVE_CYCLES %.3f" % (c, m, m / wc))
PY
