OUT=gpurun_out/r05f; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python scripts/collect_pmc.py --out $OUT --name map_highsnr --match map_ --fetch-scale 2 -- python $PWD/scripts/micro/map_highsnr_probe.py 0.01 2>&1 | tail -70
cat $OUT/map_highsnr_kernel_stats.csv | cut -c1-160 | head -8
