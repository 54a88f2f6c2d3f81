export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r06a
mkdir -p $OUT
G="SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,SQ_INSTS_BRANCH,SQ_INSTS_SMEM,SQ_INST_CYCLES_VMEM_RD,SQ_INST_CYCLES_VMEM_WR;TCP_TOTAL_CACHE_ACCESSES_sum,TCP_TCC_READ_REQ_sum,TCP_TCC_WRITE_REQ_sum,TCP_PENDING_STALL_CYCLES_sum;TA_BUSY_avr,TA_ADDR_STALLED_BY_TC_CYCLES_sum,TA_BUFFER_TOTAL_CYCLES_sum,TCP_TCP_TA_DATA_STALL_CYCLES_sum;TCP_TCC_READ_REQ_LATENCY_sum,TCC_HIT_sum,TCC_MISS_sum,TCC_REQ_sum"
CPX_LIB_PATH=$PWD/ab/libcommpy_r05.so timeout 1500 python scripts/collect_pmc.py --out $OUT --name t8mem_r05 --match turbo_pass --groups "$G" -- python $PWD/benchmarks/bench_kernels.py --which turbo8 2>&1 | tail -2
timeout 1500 python scripts/collect_pmc.py --out $OUT --name t8mem_new --match turbo_pass --groups "$G" -- python $PWD/benchmarks/bench_kernels.py --which turbo8 2>&1 | tail -2
