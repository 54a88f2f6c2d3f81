export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_devicelink_gpu.py -m gpu -q -x --timeout 300 2>&1 | tail -4
