OUT=gpurun_out/r05e; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout 180 --timeout-method=thread -k "abnormal or map or turbo or bcjr or general or fuzz" 2>&1 | tail -6 | tee $OUT/pytest_sel.txt
timeout 300 python scripts/micro/map_highsnr_probe.py 2>&1 | tee $OUT/map_highsnr_probe.txt
timeout 300 python scripts/micro/sclk_probe_check.py 2>&1 | tee $OUT/sclk_probe_check.txt
timeout 300 python benchmarks/bench_kernels.py --which map,turbo 2>&1 | cut -c1-200
