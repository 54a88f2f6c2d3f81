export TMPDIR=/tmp
for th in 704 640 768 832 960 1024 512; do
CPX_LDPC_THREADS=$th timeout 600 python benchmarks/other_configs.py --which config4 --steps 10 --warmup 3 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    j=json.loads(l)
    if 'ldpc' in j.get('kernel',''): print('threads=$th', j['kernel'][:70], '| ms', round(j.get('ms'),3), '| parity', j.get('parity',{}).get('ok'), j.get('error',''))"
done
