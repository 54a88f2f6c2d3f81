#!/bin/bash
# rocprofv3 kernel trace of the LDPC benchmark; prints per-kernel duration by launch index (iteration profile)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/ldpc_trace
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/benchmarks/bench_kernels.py --which ldpc > $OUT/log.txt 2>&1
cd $R
F=$(find $OUT/kt -name '*kernel_trace.csv' | head -1)
python - "$F" <<'PY' | tee $OUT/summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(len(rows), "launches", collections.Counter(r["Kernel_Name"][:50] for r in rows))
# split into decode calls: ldpc_init marks the start
calls, cur = [], None
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "ldpc_init" in n:
        cur = []; calls.append(cur)
    if cur is not None:
        cur.append((n, d, r["Kernel_Name"]))
for ci, c in enumerate(calls):
    if ci % 4 not in (1,):      # one representative per config roughly
        pass
    tot = collections.defaultdict(float); cnt = collections.Counter()
    for n, d, full in c:
        tot[n] += d; cnt[n] += 1
    print("call", ci, "launches", len(c), "total_us %.0f" % sum(d for _, d, _ in c), {k: (cnt[k], round(v)) for k, v in tot.items()})
# iteration profile of the last MSA and SPA calls
for ci in range(len(calls)):
    c = calls[ci]
    per = collections.defaultdict(list)
    for n, d, full in c:
        per[n].append(round(d))
    if ci in (len(calls) - 1, len(calls) - 2, 1, 2):
        for k, v in per.items():
            print("call", ci, k, v[:60])
PY
rm -rf $OUT/kt
