#!/bin/bash
# round 6: Tacotron pass time (configs[2] geometry, batch varied) on the XCD-resident decoder (the default) against the split decoder
# forced with the group count it used to pick itself (16 workgroups per utterance while they fit the 256 CUs, else 8)
set -u
OUT=$PWD/gpurun_out/r06_decoder_ab; mkdir -p $OUT
run() { timeout 200 python scripts/tacotron_bench.py --steps 5 --batch $1 "${@:2}" 2>/dev/null | grep -o '"ms_per_pass": [0-9.]*' | cut -d' ' -f2 | cut -c1-6; }
{
echo "Tacotron text->mel pass, 101 tokens, 200 decoder steps, ms per pass (5 passes averaged, two runs each)"
echo "batch | tc_decoder_x_kernel (default) | tc_decoder_g_kernel (decoder_groups = G)"
for BG in "8 16" "16 16" "24 8" "32 8"; do
  set -- $BG
  echo "$1 | $(run $1) $(run $1) | G=$2: $(run $1 --decoder-groups $2) $(run $1 --decoder-groups $2)"
done
} | tee $OUT/ab.txt
