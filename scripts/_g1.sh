OUT=gpurun_out/r05a; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout 180 --timeout-method=thread 2>&1 | tail -25 | tee $OUT/pytest_gpu.txt
BENCH_DEBUG=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err; tail -c 600 $OUT/bench_n1.err
timeout 300 python scripts/spa_rows_table.py --out $OUT > $OUT/spa_rows.txt 2>&1; tail -5 $OUT/spa_rows.txt
timeout 300 python benchmarks/bench_kernels.py --which turbo,map,viterbi_small 2>&1 | tee $OUT/bench_kernels_turbo.jsonl | cut -c1-300
