#!/bin/bash
# rocprofv3 evidence for the generation kernel (run on the GPU box: `gpurun -- bash scripts/profile_generation.sh <tag>`).
# Counters are collected in their own passes with --kernel-trace only (FETCH_SIZE and WRITE_SIZE do not fit one pass).
# Outputs land in gpurun_out/prof_<tag>/; scripts/pmc_to_traffic.py turns the two TCC passes into profiles/traffic.json.
set -u
TAG=${1:-r04}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --no-sweep --no-tacotron --no-train"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH --seconds 1.0 --steps 3 --warmup 1 > $OUT/stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -- $BENCH --seconds 0.5 --steps 1 --warmup 0 > $OUT/pmc_$C.log 2>&1
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_SQ -- $BENCH --seconds 0.5 --steps 1 --warmup 0 > $OUT/pmc_SQ.log 2>&1
# the many-streams kernel (batch 64): kernel stats of 24 000-step launches, FETCH / WRITE of a 12 000-step launch
MANY="python $REPO/scripts/many_bench.py --batch 64"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_many -- $MANY --seconds 1.0 --steps 3 --warmup 1 > $OUT/stats_many.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_many_$C -- $MANY --seconds 0.5 --steps 1 --warmup 0 > $OUT/pmc_many_$C.log 2>&1
done
# the one-hot mu-law-256 model on the XCD kernel (round 4): kernel stats of a 12 000-step launch, FETCH / WRITE of the same
MULAW="python $REPO/scripts/mulaw_bench.py --batch 8 --steps 12000 --check 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_mulaw -- $MULAW > $OUT/stats_mulaw.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_mulaw_$C -- $MULAW > $OUT/pmc_mulaw_$C.log 2>&1
done
cd $REPO
python scripts/pmc_to_traffic.py $OUT $TAG
