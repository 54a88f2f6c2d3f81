export TMPDIR=/tmp
for lib in default ab/v_front_lb4.so ab/v_front_lb5.so ab/v_front_lb6.so default ab/v_front_lb4.so ab/v_front_lb5.so ab/v_front_lb6.so; do
if [ "$lib" == "default" ]; then unset CPX_LIB_PATH; else export CPX_LIB_PATH=$PWD/$lib; fi
timeout 600 python benchmarks/other_configs.py --which config5 --steps 20 --warmup 5 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    j=json.loads(l)
    print('$lib', '| ms', round(j.get('ms'),4), '| parity', j.get('parity',{}).get('ok'), j.get('error',''), [round(v,3) for v in j.get('stage_ms',{}).values()])"
done
