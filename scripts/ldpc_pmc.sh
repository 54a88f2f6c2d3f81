#!/bin/bash
# PMC pass over the LDPC benchmark: HBM bytes and L2 hit/miss per kernel (first full iterations only)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/ldpc_pmc
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for C in "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$N -- python $R/benchmarks/bench_kernels.py --which ldpc --scale ${SCALE:-1.0} > $OUT/$N.log 2>&1
  F=$(find $OUT/$N -name '*counter_collection.csv' | head -1)
  python - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# first 12 dispatches of each kernel name = full-occupancy iterations of the first decode call
seen = collections.Counter(); acc = collections.defaultdict(lambda: collections.defaultdict(list))
disp = {}
for r in rows:
    key = (r["Dispatch_Id"], r["Counter_Name"])
    disp.setdefault(r["Dispatch_Id"], r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0])
per = collections.defaultdict(dict)
for r in rows:
    per[int(r["Dispatch_Id"])][r["Counter_Name"]] = per[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
cnt = collections.Counter()
for d in sorted(per):
    n = disp[str(d)]
    cnt[n] += 1
    if 2 <= cnt[n] <= 6:
        print(d, n, {k: round(v) for k, v in per[d].items()})
PY
  rm -rf $OUT/$N
done
