#!/bin/bash
# round 6: generation kernel check: parity tests of the XCD kernels, step time at the bench geometry, phase profile
set -u
TAG=${1:-v1}
OUT=$PWD/gpurun_out/r06_gen_$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_wavenet_gpu.py -x -q -m gpu -k "xcd_kernel_shapes or xcd_kernel_at_bench or several_streams or xcd_kernel_priming_then or matches_generic" > $OUT/pytest.txt 2>&1
tail -4 $OUT/pytest.txt
XORACLE=0 timeout 300 python scripts/xcd_check.py > $OUT/xcd_check.txt 2>&1; grep -v amdgpu.ids $OUT/xcd_check.txt | tail -8
timeout 300 python scripts/xcd_phase_profile.py > $OUT/phase.txt 2>&1; grep -v amdgpu.ids $OUT/phase.txt | head -24
