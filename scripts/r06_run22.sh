export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 900 python -m pytest $(ls tests/test_*.py | sort -r) -m gpu -q -x --timeout 300 -p no:randomly 2>&1 | tail -6 | tee gpurun_out/r06/pytest_gpu_reversed_order.txt
