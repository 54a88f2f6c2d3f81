#!/bin/bash
# HBM traffic of the Viterbi kernels of bench.py (BASELINE config 2): separate rocprofv3 PMC passes for FETCH_SIZE and
# WRITE_SIZE (+ one SQ pass), --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes.
# usage: bash scripts/viterbi_traffic.sh [tag]   ->  gpurun_out/<tag>/viterbi_pmc_summary.txt, viterbi_c2_traffic.json
TAG=${1:-r01q}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- \
      python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_$N.log 2>&1
  find $OUT/pmc_$N -name '*counter_collection.csv' -exec cp {} $OUT/pmc_$N.csv \;
  rm -rf $OUT/pmc_$N
done
cd $R
python scripts/summarize_pmc.py $OUT | grep -i "viterbi\|^==" | tee $OUT/viterbi_pmc_summary.txt
python - "$OUT" <<'PY'
import csv, json, sys, collections
out = sys.argv[1]
tot = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    per = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open("%s/pmc_%s.csv" % (out, ctr))):
        if "viterbi" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
            k = r["Kernel_Name"].split("(")[0] if not r["Kernel_Name"].startswith("void (") else r["Kernel_Name"][:80]
            per[r["Kernel_Name"][:90]][0] += float(r["Counter_Value"]); per[r["Kernel_Name"][:90]][1] += 1
    tot[ctr] = {k: v[0] / v[1] for k, v in per.items()}
json.dump(tot, open(out + "/viterbi_traffic_raw.json", "w"), indent=1)
print(json.dumps(tot, indent=1))
PY
