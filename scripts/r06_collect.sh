#!/bin/bash
# copy the outputs of scripts/r06_profile_all.sh <tag> (merged back under gpurun_out/) into profiles/ under the round's names
set -eu
TAG=${1:-r06}
P=gpurun_out/prof_$TAG
cp gpurun_out/traffic.json profiles/traffic.json
cp $P/summary_$TAG.txt profiles/r06_rocprofv3_generation_summary.txt
cp $P/summary_mfma_$TAG.txt profiles/r06_rocprofv3_mfma_summary_tacotron_train.txt
cp $P/summary_tacotron_traffic_$TAG.txt profiles/r06_rocprofv3_tacotron_traffic.txt
cp gpurun_out/train_traffic_$TAG/summary_train_traffic_$TAG.txt profiles/r06_train_traffic.txt
cp $P/stats/runc/*_kernel_stats.csv profiles/r06_rocprofv3_kernel_stats_bench_1s.csv
cp $P/stats_tacotron/runc/*_kernel_stats.csv profiles/r06_rocprofv3_kernel_stats_tacotron_c3.csv
cp $P/stats_tacotron16/runc/*_kernel_stats.csv profiles/r06_rocprofv3_kernel_stats_tacotron_b16.csv
cp $P/stats_train/runc/*_kernel_stats.csv profiles/r06_rocprofv3_kernel_stats_train_c4.csv
cp $P/stats_many/runc/*_kernel_stats.csv profiles/r06_rocprofv3_kernel_stats_many_b64_1s.csv
cp $P/stats_mulaw/runc/*_kernel_stats.csv profiles/r06_rocprofv3_kernel_stats_mulaw_b8_12000.csv
for f in xcd_phase_profile xcd_onehot_phase_profile tacotron_decoder_phase_profile tacotron_xdec_phase_profile tacotron_decoder_ab; do grep -v amdgpu.ids gpurun_out/phase_$TAG/$f.txt > profiles/r06_$f.txt; done
