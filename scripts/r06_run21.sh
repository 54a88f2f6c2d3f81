export TMPDIR=/tmp
timeout 600 python scripts/fuzz_gpu.py --seconds 60 --seed 606 2>&1 | tail -6
timeout 300 python -m pytest tests/test_jit.py -m gpu -q -x 2>&1 | tail -2
