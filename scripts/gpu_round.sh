#!/bin/bash
# One gpurun call: GPU tests, smoke, bench, rocprofv3 kernel stats + PMC passes.  Results -> gpurun_out/
# usage: bash scripts/gpu_round.sh [tag] [sections]   sections: any of t(ests) s(moke) b(ench) p(rofile) c(ounters) d(ist)
TAG=${1:-r01}
SEC=${2:-tsbpcd}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
if [[ $SEC == *t* ]]; then
  timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee $OUT/pytest_gpu.txt
fi
if [[ $SEC == *s* ]]; then
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $OUT/smoke.txt
fi
if [[ $SEC == *b* ]]; then
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 2>&1 | tail -3 | tee $OUT/bench_n1.json
  timeout 1200 python benchmarks/bench_kernels.py 2>&1 | tee $OUT/bench_kernels.jsonl | cut -c1-220
  timeout 600 python benchmarks/bench_link.py --mcs 5 2>&1 | tail -1 | tee $OUT/bench_link.jsonl | cut -c1-300
  timeout 600 python benchmarks/bench_link.py --mcs 5 --generators decimal --bits 2e7 2>&1 | tail -1 >> $OUT/bench_link.jsonl
fi
if [[ $SEC == *d* ]]; then
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
      bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -5 | tee $OUT/bench_torchrun_n1.txt
fi
if [[ $SEC == *p* ]]; then
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- \
      python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/prof_stats.log 2>&1
  cd $R
  find $OUT/prof_stats -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
  head -20 $OUT/kernel_stats.csv
fi
if [[ $SEC == *c* ]]; then
  cd /tmp
  for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-40)
    timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- \
        python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_$N.log 2>&1
    find $OUT/pmc_$N -name '*counter_collection.csv' -exec cp {} $OUT/pmc_$N.csv \;
  done
  cd $R
  python scripts/summarize_pmc.py $OUT | tee $OUT/pmc_summary.txt
  # raw traces are large: keep only the summaries
  rm -rf $OUT/pmc_*/ $OUT/prof_stats/
fi
ls -la $OUT
