export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 1500 python scripts/fuzz_gpu.py --seconds 900 --seed 20260930 2>&1 | tail -6 | tee gpurun_out/r06/fuzz_long.txt
