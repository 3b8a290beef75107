#!/bin/bash
# round 6: the XCD-resident Tacotron decoder (decoder_groups = 32): parity test, phase profile, pass time against the split kernel
set -u
TAG=${1:-v1}
OUT=$PWD/gpurun_out/r06_xdec_$TAG; mkdir -p $OUT
timeout 600 python -m pytest tests/test_tacotron_gpu.py -x -q -m gpu -k "xcd_local or bench_geometry or placement" > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
timeout 300 python scripts/tacotron_xdec_profile.py > $OUT/xdec_profile.txt 2>&1
cat $OUT/xdec_profile.txt | tail -14
for B in 32 16 8; do
  timeout 300 python scripts/tacotron_bench.py --steps 5 --batch $B > $OUT/bench_default_b$B.json 2>&1
  timeout 300 python scripts/tacotron_bench.py --steps 5 --batch $B --decoder-groups 32 > $OUT/bench_x_b$B.json 2>&1
  grep -h ms_per_pass $OUT/bench_default_b$B.json $OUT/bench_x_b$B.json | cut -c1-130
done
