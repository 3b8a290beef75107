#!/bin/bash
# round 6: pass time of the XCD-resident decoder at B = 32 / 16 against the sleeps in front of a gather's first poll (one varied at a time)
set -u
OUT=$PWD/gpurun_out/r06_napsweep; mkdir -p $OUT
run() { timeout 200 python scripts/tacotron_bench.py --steps 5 --batch $1 "${@:2}" 2>/dev/null | grep -o '"ms_per_pass": [0-9.]*'; }
for B in 32 16; do
  echo "B=$B default: $(run $B) $(run $B)"
  for v in 0 4 16 24; do echo "B=$B nap-idle $v: $(run $B --nap-idle $v)"; done
  for v in 0 4 8 16 20; do echo "B=$B nap-owner $v: $(run $B --nap-owner $v)"; done
  for v in 0 1 6 10; do echo "B=$B nap-w0 $v: $(run $B --nap-w0 $v)"; done
  for v in 0 2 4; do echo "B=$B nap-round $v: $(run $B --nap-round $v)"; done
done | tee $OUT/sweep.txt
