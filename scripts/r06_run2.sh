export TMPDIR=/tmp
mkdir -p gpurun_out/r06a
timeout 300 python scripts/micro/turbo_tm_debug.py 2>&1 | tail -20 | tee gpurun_out/r06a/debug.txt
bash scripts/ab_kernels.sh r06a turbo 2 ab/libcommpy_r05.so default 2>&1 | tail -20
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r06a/prof -o turbo -- python $GRAFT_REPO_ROOT/benchmarks/bench_kernels.py --which turbo > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/r06a/prof -name "*kernel_stats*" | head -1 | xargs head -12
