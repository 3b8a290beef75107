#!/bin/bash
# HBM traffic of the training step's kernels (`gpurun -- bash scripts/train_traffic.sh [tag]`): FETCH_SIZE and WRITE_SIZE in
# separate passes (KiB; FETCH_SIZE x 2 on gfx950, MI355X_MICROARCH.md), printed per launch next to the kernel's average duration;
# the whole step's bytes go to profiles/traffic.json["train:" + _lib.train_hash()] (bench.py's train.roofline.traffic) and the
# table to gpurun_out/train_traffic_<tag>/summary_train_traffic_<tag>.txt.
set -u
TAG=${1:-quick}
REPO=$PWD
OUT=$PWD/gpurun_out/train_traffic_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -- python $REPO/scripts/train_bench.py --steps 1 --warmup 1 > $OUT/pmc_$C.log 2>&1
done
cd $REPO
python scripts/pmc_to_train_traffic.py "$OUT" "$TAG"
mkdir -p $REPO/gpurun_out && cp $REPO/profiles/traffic.json $REPO/gpurun_out/traffic.json
