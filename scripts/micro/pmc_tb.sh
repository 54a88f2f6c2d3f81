#!/bin/bash
# PMC passes over bench.py restricted to the Viterbi codeword-path kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_tb
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/p$i.log 2>&1
  find $OUT/p$i -name '*counter_collection.csv' -exec cp {} $OUT/p$i.csv \;
  rm -rf $OUT/p$i
done
python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$OUT/p*.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "viterbi" not in k: continue
        k = "acs" if "acs" in k else "tb"
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        print(k, {c: "%.4g" % (v / cnt[(k, c)]) for c, v in d.items()})
PY
