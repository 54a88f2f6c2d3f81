export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_bcjr_ldpc_demod_gpu.py tests/test_general_gpu.py tests/test_wifi_gpu.py tests/test_devicelink_gpu.py tests/test_fp32_fast_gpu.py -m gpu -q -x --timeout 300 -k "demod or wifi or link or modem" 2>&1 | tail -8
timeout 600 python benchmarks/bench_kernels.py --which demod 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    j=json.loads(l); print('%-46s %-64s %8.4f ms  %5.1f %% of HBM' % (j['kernel'][:46], j['workload'][:64], j['ms'], 100*j['roofline']['frac']))"
timeout 900 python benchmarks/other_configs.py --steps 20 --warmup 5 2>&1 | grep "^{" | tee gpurun_out/r06/bench_other_configs.jsonl | python -c "
import sys, json
for l in sys.stdin:
    j=json.loads(l)
    print(j.get('config'), '|', j.get('kernel','')[:60], '| ms', j.get('ms'), '| parity', j.get('parity',{}).get('ok'), j.get('error',''), j.get('stage_ms',''))"
