export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_viterbi_cw_gpu.py tests/test_config_sizes_gpu.py tests/test_devicelink_gpu.py tests/test_wifi_gpu.py tests/test_abnormal_golden_gpu.py -m gpu -q -x --timeout 300 2>&1 | tail -5
for ov in 0 1 0 1; do
CPX_VITERBI_OVERLAP=$ov timeout 600 python benchmarks/other_configs.py --which config5 --steps 20 --warmup 5 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    j=json.loads(l)
    print('overlap=$ov', j.get('kernel','')[-90:], '| ms', round(j.get('ms'),4), '| parity', j.get('parity',{}).get('ok'), j.get('error',''), [round(v,3) for v in j.get('stage_ms',{}).values()])"
done
for ov in 0 1; do echo overlap=$ov; CPX_VITERBI_OVERLAP=$ov python scripts/micro/split_probe.py 2>&1 | tail -6; done
