export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r06a
mkdir -p $OUT
timeout 900 python -m pytest tests/test_bcjr_ldpc_demod_gpu.py tests/test_abnormal_golden_gpu.py tests/test_config_sizes_gpu.py tests/test_fp32_fast_gpu.py tests/test_general_gpu.py tests/test_large_sizes_gpu.py tests/test_encoders_gpu.py tests/test_fuzz_slice_gpu.py tests/test_round2_gpu.py -m gpu -q -x --timeout 300 -k "turbo or map" 2>&1 | tail -3
timeout 1500 python scripts/collect_pmc.py --out $OUT --name turbo_c3 --match turbo_ --fetch-scale 2 -- python $PWD/benchmarks/bench_kernels.py --which turbo 2>&1 | tail -30
timeout 1500 python scripts/collect_pmc.py --out $OUT --name turbo8 --match turbo_ --fetch-scale 2 -- python $PWD/benchmarks/bench_kernels.py --which turbo8 2>&1 | tail -30
