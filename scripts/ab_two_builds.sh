#!/bin/bash
# A/B of two builds of libtwv_amd.so ON THE SAME BOX, alternating (box-to-box differences on this pool are 0.5-1 %, more than most
# single edits are worth).  Prepare here:   build A;  cp <pkg>/libtwv_amd.so <pkg>/libtwv_A.so.keep ;  build B;  cp ... libtwv_B.so.keep
# then on the GPU box:   bash scripts/ab_two_builds.sh [reps] -- <command printing one number per run>
# (the library is loaded as it is found: _lib.lib() does not rebuild).  Restores B at the end.
set -u
D=tacotron-wavenet-vocoder-korean_amd
REPS=3
if [ "${1:-}" != "--" ]; then REPS=$1; shift; fi
shift
for rep in $(seq $REPS); do
  for v in A B; do
    cp $D/libtwv_$v.so.keep $D/libtwv_amd.so
    echo "$v  $("$@" 2>/dev/null | tr '\n' ' ')"
  done
done
