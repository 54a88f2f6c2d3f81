#!/bin/bash
# One gpurun call: GPU tests, smoke, bench, rocprofv3 kernel stats + PMC passes.  Results -> gpurun_out/<tag>/
# usage: bash scripts/gpu_round.sh [tag] [sections]
#   sections: t(ests) s(moke) b(ench.py) d(ist: torchrun N=1 over the engine's RCCL binding) k(ernel benches)
#             l(ink + host-api + multi-GPU benches) v(iterbi PMC passes) u(turbo/map PMC passes) m(demod PMC passes)
#             x(ldpc PMC passes) r(octx marker trace of the host-API benchmark, CPX_TRACE=1)
#             c(alibration of FETCH_SIZE / WRITE_SIZE on known byte counts, scripts/micro/fetch_calib.py)
#             f(uzz: scripts/fuzz_gpu.py for 90 s) L(ong run: bench.py --steps 500) T(olerance table of the sum-product decoder)
#             o(ther_configs on their own) h(igh-SNR map_decode probe + PMC of the literal kernel) R(ow-vs-row sum-product table)
#             P(robe check: does the shader-clock probe disturb what it measures) O(ther_configs PMC passes, one workload at a time)
TAG=${1:-r05}
SEC=${2:-tsbdklvu}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
if [[ $SEC == *t* ]]; then
  timeout 900 python -m pytest tests -m gpu -q -x --timeout 120 --timeout-method=thread --durations=8 2>&1 | tail -120 | tee $OUT/pytest_gpu.txt
fi
if [[ $SEC == *s* ]]; then
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $OUT/smoke.txt
fi
if [[ $SEC == *b* ]]; then
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | tail -3 | tee $OUT/bench_n1.json
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --precision fp32-fast 2>&1 | tail -1 | tee $OUT/bench_fp32_fast.json | cut -c1-400
fi
if [[ $SEC == *d* ]]; then
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
      bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --gather 2>&1 | tail -5 | tee $OUT/bench_torchrun_n1.txt
fi
if [[ $SEC == *k* ]]; then
  timeout 1200 python benchmarks/bench_kernels.py 2>&1 | tee $OUT/bench_kernels.jsonl | cut -c1-250
  # the fp32-fast precision mode (not the parity mode): the kernels that have a float32 variant
  CPX_PRECISION=fp32-fast timeout 600 python benchmarks/bench_kernels.py --which ldpc,config4,demod,turbo,turbo8 2>&1 | grep "^{" | tee $OUT/bench_kernels_fp32_fast.jsonl | cut -c1-200
fi
if [[ $SEC == *l* ]]; then
  timeout 600 python benchmarks/bench_link.py --mcs 5 2>&1 | tail -1 | tee $OUT/bench_link.jsonl | cut -c1-300
  timeout 600 python benchmarks/bench_link.py --mcs 5 --generators decimal 2>&1 | tail -1 | tee -a $OUT/bench_link.jsonl | cut -c1-300
  timeout 600 python benchmarks/bench_host_api.py 2>&1 | tail -2 | tee $OUT/bench_host_api.json | cut -c1-300
  timeout 900 python benchmarks/bench_multigpu.py 2>&1 | grep "^{" | tee $OUT/bench_multigpu.jsonl | cut -c1-330
fi
if [[ $SEC == *v* ]]; then
  timeout 1500 python scripts/collect_pmc.py --out $OUT --name viterbi_c2 --match viterbi --batch 65536 -- \
      python $R/bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-other-configs --sustain-seconds 0 2>&1 | tail -40
fi
if [[ $SEC == *u* ]]; then
  timeout 1500 python scripts/collect_pmc.py --out $OUT --name turbo_c3 --match _kernel --fetch-scale 2 -- \
      python $R/benchmarks/bench_kernels.py --which turbo,map 2>&1 | tail -60
fi
if [[ $SEC == *m* ]]; then
  timeout 1500 python scripts/collect_pmc.py --out $OUT --name demod --match demod_ --fetch-scale 2 -- \
      python $R/benchmarks/bench_kernels.py --which demod 2>&1 | tail -60
fi
if [[ $SEC == *x* ]]; then
  timeout 1500 python scripts/collect_pmc.py --out $OUT --name ldpc_c4 --match ldpc_ --fetch-scale 2 -- \
      python $R/benchmarks/bench_kernels.py --which ldpc 2>&1 | tail -80
  timeout 900 python scripts/collect_pmc.py --out $OUT --name ldpc_resident_fixed20 --match ldpc_resident --fetch-scale 2 -- \
      python $R/scripts/micro/ldpc_fixed_iters.py 2>&1 | tail -5
fi
if [[ $SEC == *T* ]]; then
  timeout 600 python scripts/spa_tolerance_table.py --out $OUT 2>&1 | tail -3
fi
if [[ $SEC == *o* ]]; then
  timeout 600 python benchmarks/other_configs.py --steps 20 --warmup 5 2>&1 | grep "^{" | tee $OUT/bench_other_configs.jsonl | cut -c1-260
fi
if [[ $SEC == *O* ]]; then
  # counters for what the driver times (round-5 verdict item 4): PMC passes over benchmarks/other_configs.py itself, one workload at a time
  for w in turbo config4 config5 config1 pair; do
    timeout 1500 python scripts/collect_pmc.py --out $OUT --name oc_$w --match _kernel --fetch-scale 2 -- \
        python $R/benchmarks/other_configs.py --which $w --steps 10 --warmup 3 2>&1 | tail -3 | cut -c1-300
  done
fi
if [[ $SEC == *h* ]]; then
  timeout 300 python scripts/micro/map_highsnr_probe.py 2>&1 | tee $OUT/map_highsnr_probe.txt
  timeout 900 python scripts/collect_pmc.py --out $OUT --name map_highsnr --match map_ --fetch-scale 2 -- \
      python $R/scripts/micro/map_highsnr_probe.py 0.01 2>&1 | tail -5
fi
if [[ $SEC == *R* ]]; then
  timeout 300 python scripts/spa_rows_table.py --out $OUT > $OUT/spa_rows.txt 2>&1; tail -2 $OUT/spa_rows.txt | cut -c1-200
fi
if [[ $SEC == *P* ]]; then
  timeout 300 python scripts/micro/sclk_probe_check.py 2>&1 | tee $OUT/sclk_probe_check.txt
fi
if [[ $SEC == *c* ]]; then
  timeout 600 python scripts/micro/fetch_calib.py --out $OUT 2>&1 | tail -3
fi
if [[ $SEC == *f* ]]; then
  timeout 300 python scripts/fuzz_gpu.py --seconds 90 2>&1 | tail -4 | tee $OUT/fuzz.txt
fi
if [[ $SEC == *L* ]]; then
  timeout 600 python bench.py --gpus 1 --steps 500 --warmup 5 --no-cpu-baseline --sustain-seconds 0 2>&1 | tail -1 | tee $OUT/bench_n1_steps500.json | cut -c1-300
fi
if [[ $SEC == *r* ]]; then
  # roctx ranges of the entry points (CPX_TRACE=1) next to the kernels: rocprofv3 marker trace of the host-API benchmark
  rm -rf /tmp/cpx_marker
  (cd /tmp && CPX_TRACE=1 timeout 600 rocprofv3 --marker-trace --kernel-trace --output-format csv -d /tmp/cpx_marker -- \
      python $R/benchmarks/bench_host_api.py --reps 1 > $OUT/roctx_marker.log 2>&1)
  f=$(find /tmp/cpx_marker -name "*marker_api_trace.csv" | head -1)
  if [ -n "$f" ]; then head -40 "$f" > $OUT/roctx_marker_trace_head.csv; wc -l "$f" | tee -a $OUT/roctx_marker_trace_head.csv; fi
fi
rm -f $OUT/*.log
ls -la $OUT
