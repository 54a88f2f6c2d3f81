export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_devicelink_gpu.py tests/test_wifi_gpu.py tests/test_bcjr_ldpc_demod_gpu.py -m gpu -q -x --timeout 300 2>&1 | tail -15
timeout 600 python benchmarks/other_configs.py --which config5 --steps 20 --warmup 5 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    j=json.loads(l)
    print(j.get('config'), '|', j.get('kernel','')[:100], '| ms', j.get('ms'), '| parity', j.get('parity'), j.get('error',''), j.get('stage_ms',''))"
timeout 600 python benchmarks/bench_kernels.py --which demod 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    j=json.loads(l); print('%-46s %-64s %8.4f ms  %5.1f %% of HBM' % (j['kernel'][:46], j['workload'][:64], j['ms'], 100*j['roofline']['frac']))"
