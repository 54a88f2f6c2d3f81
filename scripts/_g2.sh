OUT=gpurun_out/r05b; mkdir -p $OUT; export TMPDIR=/tmp
for f in 1 2 0; do
  echo "== CPX_TURBO_FOLD=$f"
  CPX_TURBO_FOLD=$f timeout 600 python -m pytest tests -m gpu -q -x --timeout 180 -k "turbo or map or abnormal or bcjr" 2>&1 | tail -3
  for r in 1 2; do CPX_TURBO_FOLD=$f timeout 300 python benchmarks/bench_kernels.py --which turbo,turbo8 2>&1 | grep turbo_decode | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  fold=$f', d['workload'][:40], round(d['ms'], 3), 'ms  ber', d.get('ber'))"; done
done 2>&1 | tee $OUT/turbo_fold_ab.txt
