python -m pytest tests/test_tacotron_gpu.py -x -q 2>&1 | tail -2
for r in 1 2; do python scripts/tacotron_bench.py --steps 10 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_pass'])"; done
python scripts/tacotron_phase_profile.py 2>&1 | tail -9
