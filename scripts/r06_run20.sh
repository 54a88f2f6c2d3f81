export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_jit.py tests/test_viterbi_cw_gpu.py tests/test_random_codes_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -8
timeout 600 python benchmarks/bench_kernels.py --which viterbi_variants 2>&1 | grep "^{\|^#" | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('#'): print(l.strip()); continue
    j=json.loads(l); print('%-80s %-70s %8.4f ms' % (j['kernel'][:80], j['workload'][:70], j['ms']))"
