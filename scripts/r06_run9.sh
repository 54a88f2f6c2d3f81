export TMPDIR=/tmp
mkdir -p gpurun_out/r06a
python scripts/micro/turbo_slot_probe.py 2>&1 | tail -3
timeout 900 python -m pytest tests/test_bcjr_ldpc_demod_gpu.py tests/test_abnormal_golden_gpu.py tests/test_config_sizes_gpu.py tests/test_fp32_fast_gpu.py tests/test_general_gpu.py tests/test_large_sizes_gpu.py tests/test_encoders_gpu.py tests/test_fuzz_slice_gpu.py tests/test_round2_gpu.py -m gpu -q -x --timeout 300 -k "turbo or map" 2>&1 | tail -5 | tee gpurun_out/r06a/pytest_turbo.txt
rm -f gpurun_out/r06a/ab.jsonl
bash scripts/ab_kernels.sh r06a turbo,turbo8 3 ab/libcommpy_r05.so ab/v_b128.so default 2>&1 | grep -v map_decode | tail -30
