#!/bin/bash
# LDS bank-conflict survey of every workload's kernels (tuning aid): SQ_LDS_BANK_CONFLICT (cycles LDS is stalled by bank conflicts),
# SQ_LDS_ADDR_CONFLICT, SQ_LDS_IDX_ACTIVE (cycles the LDS index unit is busy), SQ_INSTS_LDS, SQ_BUSY_CYCLES -- one rocprofv3 --pmc pass each
set -u
TAG=${1:-r04}
REPO=$PWD
OUT=$PWD/gpurun_out/lds_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
declare -A CMD
CMD[gen]="python $REPO/bench.py --no-cpu-baseline --no-sweep --no-tacotron --no-train --seconds 0.3 --steps 1 --warmup 0"
CMD[mulaw]="python $REPO/scripts/mulaw_bench.py --batch 8 --steps 4000 --check 0"
CMD[tacotron]="python $REPO/scripts/tacotron_bench.py --steps 2"
CMD[train]="python $REPO/scripts/train_bench.py --steps 2 --warmup 1"
for W in gen mulaw tacotron train; do
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/$W -- ${CMD[$W]} > $OUT/$W.log 2>&1
done
cd $REPO
python - "$OUT" <<'PY'
import csv, glob, sys, collections
for w in ("gen", "mulaw", "tacotron", "train"):
    fs = glob.glob(sys.argv[1] + "/" + w + "/**/*counter_collection.csv", recursive=True)
    if not fs: print(w, "no counters"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs[0])):
        acc[r["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0][:48]][r["Counter_Name"]] += float(r["Counter_Value"])
    print("==", w)
    for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0)):
        idx = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
        if idx < 1e5: continue
        print("  %-48s LDS_IDX_ACTIVE %.3e  BANK_CONFLICT %.3e (%.0f %% of active)  ADDR_CONFLICT %.3e  INSTS_LDS %.3e  GUI_ACTIVE %.3e" % (
            k, idx, c.get("SQ_LDS_BANK_CONFLICT", 0), 100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / idx, c.get("SQ_LDS_ADDR_CONFLICT", 0), c.get("SQ_INSTS_LDS", 0), c.get("GRBM_GUI_ACTIVE", 0)))
PY
