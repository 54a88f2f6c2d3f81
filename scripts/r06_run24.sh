export TMPDIR=/tmp
CPX_VITERBI_OVERLAP=1 timeout 900 python -m pytest tests/test_viterbi_cw_gpu.py tests/test_config_sizes_gpu.py tests/test_devicelink_gpu.py tests/test_wifi_gpu.py tests/test_abnormal_golden_gpu.py -m gpu -q -x --timeout 300 2>&1 | tail -3
for ov in 0 1 0 1; do
CPX_VITERBI_OVERLAP=$ov timeout 600 python benchmarks/other_configs.py --which config5 --steps 20 --warmup 5 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    j=json.loads(l)
    print('overlap=$ov', j.get('kernel','')[:150], '| ms', round(j.get('ms'),4), '| parity', j.get('parity',{}).get('ok'), j.get('error',''), [round(v,3) for v in j.get('stage_ms',{}).values()])"
done
for ov in 0 1; do echo overlap=$ov; CPX_VITERBI_OVERLAP=$ov python scripts/micro/split_probe.py 2>&1 | tail -2; done
cd /tmp
rm -rf /tmp/kt1
CPX_VITERBI_OVERLAP=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt1 -- python $GRAFT_REPO_ROOT/benchmarks/other_configs.py --which config5 --steps 3 --warmup 2 > /dev/null 2>&1
f=$(find /tmp/kt1 -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'viterbi' in r['Kernel_Name']]
rows=rows[-4:]
t0=int(rows[0]['Start_Timestamp'])
for r in rows: print(r['Kernel_Name'][40:110], 'queue', r.get('Queue_Id'), 'start', (int(r['Start_Timestamp'])-t0)/1e3, 'end', (int(r['End_Timestamp'])-t0)/1e3)
PY
