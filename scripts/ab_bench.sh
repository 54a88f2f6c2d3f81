#!/bin/bash
# Same-box A/B of bench.py over several builds of the engine (CPX_LIB_PATH), interleaved so that clock drift hits all arms alike.
# usage: bash scripts/ab_bench.sh <tag> <rounds> lib1.so lib2.so ...   ("default" = the in-tree library)
TAG=$1; ROUNDS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for r in $(seq 1 $ROUNDS); do
  for lib in "$@"; do
    if [ "$lib" == "default" ]; then unset CPX_LIB_PATH; else export CPX_LIB_PATH=$R/$lib; fi
    timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print(json.dumps({'lib': '$lib', 'round': $r, 'ms_per_step': round(j['ms_per_step'], 4), 'kernel_ms_avg': round(j['roofline']['kernel_ms_avg'], 4), 'mismatch': j['oracle_mismatched_bits'], 'ber': j['ber']}))" | tee -a $OUT/ab.jsonl
  done
done
