#!/bin/bash
# round 6 evidence in one GPU call: bash scripts/r06_profile_all.sh <tag>
#   generation: kernel stats + FETCH / WRITE / SQ passes (scripts/profile_generation.sh) -> traffic.json[generation hash]
#   Tacotron + training: kernel stats, MFMA counters, FETCH / WRITE passes (scripts/profile_secondaries.sh) -> traffic.json["tacotron:..."]
#   training step traffic (scripts/train_traffic.sh) -> traffic.json["train:..."]
#   phase profiles (generation MoL / one-hot, Tacotron split decoder, XCD-resident decoder)
# everything lands under gpurun_out/ (merged back), incl. the refreshed profiles/traffic.json as gpurun_out/traffic.json
set -u
TAG=${1:-r06}
REPO=$PWD
bash scripts/profile_generation.sh $TAG > gpurun_out/profile_generation_$TAG.log 2>&1
bash scripts/profile_secondaries.sh $TAG > gpurun_out/profile_secondaries_$TAG.log 2>&1
bash scripts/train_traffic.sh $TAG > gpurun_out/train_traffic_$TAG.log 2>&1
cd $REPO
mkdir -p gpurun_out/phase_$TAG
python scripts/xcd_phase_profile.py > gpurun_out/phase_$TAG/xcd_phase_profile.txt 2>&1
python scripts/xcd_phase_profile.py --onehot > gpurun_out/phase_$TAG/xcd_onehot_phase_profile.txt 2>&1
python scripts/tacotron_phase_profile.py > gpurun_out/phase_$TAG/tacotron_decoder_phase_profile.txt 2>&1
python scripts/tacotron_xdec_profile.py > gpurun_out/phase_$TAG/tacotron_xdec_phase_profile.txt 2>&1
bash scripts/r06_decoder_ab.sh > /dev/null 2>&1
cp gpurun_out/r06_decoder_ab/ab.txt gpurun_out/phase_$TAG/tacotron_decoder_ab.txt
cp profiles/traffic.json gpurun_out/traffic.json
# keep the merged-back volume small: the raw counter CSVs of the training / Tacotron MFMA passes are tens of MB
find gpurun_out/prof_$TAG -name "*.db" -delete 2>/dev/null
du -sh gpurun_out/prof_$TAG gpurun_out/train_traffic_$TAG 2>/dev/null
