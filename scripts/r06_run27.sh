export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bcjr_ldpc_demod_gpu.py tests/test_large_sizes_gpu.py tests/test_abnormal_golden_gpu.py tests/test_config_sizes_gpu.py tests/test_fp32_fast_gpu.py -m gpu -q -x --timeout 300 -k "turbo or map or abnormal or config3" 2>&1 | tail -3
for rw in 1 0 1 0; do
CPX_TURBO_RW16=$rw timeout 600 python benchmarks/bench_kernels.py --which turbo8,turbo 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    j=json.loads(l); print('rw16=$rw %-50s %-60s %8.4f ms' % (j['kernel'][:50], j['workload'][:60], j['ms']))"
done
