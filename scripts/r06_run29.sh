export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_viterbi_cw_gpu.py tests/test_config_sizes_gpu.py tests/test_devicelink_gpu.py tests/test_random_codes_gpu.py tests/test_viterbi_gpu.py -m gpu -q -x --timeout 300 2>&1 | tail -3
for lib in ab/libcommpy_prev.so default ab/libcommpy_prev.so default; do
if [ "$lib" == "default" ]; then unset CPX_LIB_PATH; else export CPX_LIB_PATH=$PWD/$lib; fi
timeout 600 python benchmarks/other_configs.py --which config1 --steps 20 --warmup 5 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    j=json.loads(l)
    print('$lib', j['kernel'][:60], '| ms', round(j.get('ms'),4), '| parity', j.get('parity',{}).get('ok'), j.get('error',''))"
done
