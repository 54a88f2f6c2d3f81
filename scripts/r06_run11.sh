echo "== new"; python scripts/micro/turbo_perm_probe.py 2>&1 | tail -6
echo "== r05"; CPX_LIB_PATH=$PWD/ab/libcommpy_r05.so python scripts/micro/turbo_perm_probe.py 2>&1 | tail -6
