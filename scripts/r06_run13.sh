export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r06a
CPX_LIB_PATH=$PWD/ab/libcommpy_r05.so timeout 1500 python scripts/collect_pmc.py --out $OUT --name turbo8_r05 --match turbo_ --fetch-scale 2 -- python $PWD/benchmarks/bench_kernels.py --which turbo8 2>&1 | tail -3
timeout 1500 python scripts/collect_pmc.py --out $OUT --name turbo8 --match turbo_ --fetch-scale 2 -- python $PWD/benchmarks/bench_kernels.py --which turbo8 2>&1 | tail -3
