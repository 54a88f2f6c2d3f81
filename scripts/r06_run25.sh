export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_viterbi_cw_gpu.py -m gpu -q -x --timeout 300 -k "remainder or full_size" 2>&1 | tail -4
