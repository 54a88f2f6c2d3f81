python scripts/micro/turbo_slot_probe.py 2>&1 | tail -3
rm -f gpurun_out/r06a/ab.jsonl
bash scripts/ab_kernels.sh r06a turbo,turbo8 2 ab/libcommpy_r05.so default ab/v_nodec.so ab/v_noidx.so ab/v_nollr.so ab/v_none.so 2>&1 | python -c "
import sys, json, collections
d=collections.defaultdict(list)
n=collections.Counter()
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); n[(j['lib'],j['round'])]+=1
        d[(j['lib'], 'S4' if n[(j['lib'],j['round'])]==1 else 'S8')].append(j['ms'])
for k in sorted(d): print(k, ['%.3f'%x for x in d[k]])
"
