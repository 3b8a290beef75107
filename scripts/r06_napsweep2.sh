#!/bin/bash
# round 6: the wave-0 nap of the XCD-resident decoder again, three runs per value (B = 32)
set -u
OUT=$PWD/gpurun_out/r06_napsweep2; mkdir -p $OUT
run() { timeout 200 python scripts/tacotron_bench.py --steps 8 --batch 32 "$@" 2>/dev/null | grep -o '"ms_per_pass": [0-9.]*' | cut -d' ' -f2 | cut -c1-6; }
for rep in 1 2 3; do
  for v in 3 1 0 2 5; do echo "rep $rep nap-w0 $v: $(run --nap-w0 $v)"; done
  for v in 12 10 14; do echo "rep $rep nap-owner $v: $(run --nap-owner $v)"; done
done | tee $OUT/sweep.txt
