#!/bin/bash
# rocprofv3 evidence for the MFMA kernels of the two secondary rows (run on the GPU box: `gpurun -- bash scripts/profile_secondaries.sh <tag>`):
#   configs[2] Tacotron text->mel  (scripts/tacotron_bench.py: tc_gemm_mfma_kernel, tc_gemm_mfma_ck_kernel, tc_decoder_*)
#   configs[3] WaveNet training    (scripts/train_bench.py:    tr_layer_{fwd,bwd1,bwd2}_kernel + the wide GEMMs)
# Kernel stats in one run; counters in their own runs with --kernel-trace only (MI355X_MICROARCH.md: 8 SQ slots per pass).
# Outputs land in gpurun_out/prof_<tag>/; scripts/pmc_to_mfma.py condenses them into summary_mfma_<tag>.txt.
set -u
TAG=${1:-r04}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_available.txt 2>&1
for W in tacotron train; do
  if [ $W = tacotron ]; then CMD="python $REPO/scripts/tacotron_bench.py --steps 3"; else CMD="python $REPO/scripts/train_bench.py --steps 3 --warmup 1"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$W -- $CMD > $OUT/stats_$W.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma_$W -- $CMD > $OUT/pmc_mfma_$W.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_mops_$W -- $CMD > $OUT/pmc_mops_$W.log 2>&1
done
# HBM bytes of a configs[2] pass (tacotron.roofline.traffic was null): FETCH_SIZE / WRITE_SIZE, one counter per pass
# (they do not fit one pass), kernel trace only; scripts/pmc_to_tacotron_traffic.py condenses them into profiles/traffic.json
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_taco_$C -- python $REPO/scripts/tacotron_bench.py --steps 3 > $OUT/pmc_taco_$C.log 2>&1
done
# round 6: the XCD-resident decoder (weights in registers for the whole launch): forced at the bench batch, and at batch 16 where it is the default
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_tacog32_$C -- python $REPO/scripts/tacotron_bench.py --steps 3 --decoder-groups 8 > $OUT/pmc_tacog32_$C.log 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_taco16_$C -- python $REPO/scripts/tacotron_bench.py --steps 3 --batch 16 > $OUT/pmc_taco16_$C.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_tacotron16 -- python $REPO/scripts/tacotron_bench.py --steps 3 --batch 16 > $OUT/stats_tacotron16.log 2>&1
cd $REPO
python scripts/pmc_to_mfma.py $OUT $TAG
python scripts/pmc_to_tacotron_traffic.py $OUT $TAG
