#!/bin/bash
# run a subset (or all) of the GPU tests:  bash scripts/r06_tests.sh <tag> [pytest args...]
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/r06_tests_$TAG; mkdir -p $OUT
timeout 1500 python -m pytest -x -q -m gpu "$@" > $OUT/pytest.txt 2>&1
tail -15 $OUT/pytest.txt
