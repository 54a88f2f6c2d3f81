export TMPDIR=/tmp
python scripts/micro/turbo_diff_probe.py 1 /tmp/new1.npy
CPX_LIB_PATH=$PWD/ab/libcommpy_r05.so python scripts/micro/turbo_diff_probe.py 1 /tmp/old1.npy
python scripts/micro/turbo_diff_probe.py --diff /tmp/new1.npy /tmp/old1.npy
