#!/bin/bash
# A/B: same bench, several builds of the library (TWV_AMD_LIB), interleaved
for rep in 1 2; do
for v in "$@"; do
  echo -n "$v: "
  TWV_AMD_LIB=$PWD/scripts/libtwv_$v.so.bin python bench.py --seconds 2 --steps 2 --warmup 1 --no-tacotron --no-train --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['roofline']['us_per_generation_step'], d['value'])"
done; done
