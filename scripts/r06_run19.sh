export TMPDIR=/tmp
mkdir -p gpurun_out/r06
for ov in 0 1 2; do
CPX_VITERBI_OVERLAP=$ov timeout 600 python benchmarks/other_configs.py --which config5 --steps 20 --warmup 5 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    j=json.loads(l)
    print('overlap=$ov', j.get('kernel','')[-90:], '| ms', round(j.get('ms'),4), '| parity', j.get('parity',{}).get('ok'), j.get('error',''), [round(v,3) for v in j.get('stage_ms',{}).values()])"
done
cd /tmp
for ov in 1 2; do
rm -rf /tmp/kt$ov
CPX_VITERBI_OVERLAP=$ov rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$ov -- python $GRAFT_REPO_ROOT/benchmarks/other_configs.py --which config5 --steps 3 --warmup 2 > /dev/null 2>&1
f=$(find /tmp/kt$ov -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'viterbi' in r['Kernel_Name']]
rows=rows[-6:]
t0=int(rows[0]['Start_Timestamp'])
for r in rows: print(r['Kernel_Name'][:60], 'queue', r.get('Queue_Id'), 'start', (int(r['Start_Timestamp'])-t0)/1e3, 'end', (int(r['End_Timestamp'])-t0)/1e3)
PY
done
