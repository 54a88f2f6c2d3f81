export TMPDIR=/tmp
mkdir -p gpurun_out/r06a
echo "== new"; timeout 300 python scripts/micro/turbo_flag_probe.py 2>&1 | tail -10 | tee gpurun_out/r06a/flags_new.txt
echo "== r05"; CPX_LIB_PATH=$PWD/ab/libcommpy_r05.so timeout 300 python scripts/micro/turbo_flag_probe.py 2>&1 | tail -10 | tee gpurun_out/r06a/flags_r05.txt
timeout 600 python -m pytest tests/test_fp32_fast_gpu.py -m gpu -q -x --timeout 300 -k "turbo" 2>&1 | tail -5
