#!/bin/bash
# Same-box A/B of benchmarks/bench_kernels.py over several builds (CPX_LIB_PATH), interleaved.
# usage: bash scripts/ab_kernels.sh <tag> <which> <rounds> lib1.so lib2.so ...   ("default" = the in-tree library)
TAG=$1; WHICH=$2; ROUNDS=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for r in $(seq 1 $ROUNDS); do
  for lib in "$@"; do
    if [ "$lib" == "default" ]; then unset CPX_LIB_PATH; else export CPX_LIB_PATH=$R/$lib; fi
    timeout 300 python benchmarks/bench_kernels.py --which $WHICH 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line)
        print(json.dumps({'lib': '$lib', 'round': $r, 'kernel': j['kernel'], 'ms': round(j['ms'], 4)}))" | tee -a $OUT/ab.jsonl
  done
done
