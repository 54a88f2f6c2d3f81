OUT=gpurun_out/r05d; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout 180 --timeout-method=thread 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
timeout 300 python scripts/micro/map_highsnr_probe.py 2>&1 | tee $OUT/map_highsnr_probe.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err; tail -c 300 $OUT/bench_n1.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05d/bench_n1.json'))
print(d['value'], d['ms_per_step'], d['roofline']['clock'])
for e in d['other_configs']: print(e['config'], e.get('ms'), e.get('parity',{}).get('ok'), e.get('error'))
PY
