#!/bin/bash
# PMC comparison: production codeword-path ACS kernel vs the micro-benchmark kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_cw
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_SALU SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/prod$i -- python $R/scripts/micro/vit_cw_probe.py one > $OUT/prod$i.log 2>&1
  find $OUT/prod$i -name '*counter_collection.csv' -exec cp {} $OUT/prod$i.csv \;
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/micro$i -- $R/scripts/micro/acs_len2060 > $OUT/micro$i.log 2>&1
  find $OUT/micro$i -name '*counter_collection.csv' -exec cp {} $OUT/micro$i.csv \;
  rm -rf $OUT/prod$i $OUT/micro$i
done
python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$OUT/*.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        cnt[(k, r["Counter_Name"])] += 1
    print(f.split("/")[-1])
    for k, d in agg.items():
        if "acs" in k:
            print("  ", k, {c: round(v / cnt[(k, c)]) for c, v in d.items()})
PY
