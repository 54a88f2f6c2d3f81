for lib in default variants/lit_DIVNR.so variants/lit_NOSUM.so variants/lit_EXP1.so; do
  if [ "$lib" == "default" ]; then unset CPX_LIB_PATH; else export CPX_LIB_PATH=$PWD/$lib; fi
  echo "== $lib"; timeout 200 python scripts/micro/map_highsnr_probe.py 0.01,0.01 2>&1 | tail -2
done
