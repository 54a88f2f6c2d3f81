export TMPDIR=/tmp
for r in 1 2; do
for lib in default ab/v_ldpc_nobar_spa.so; do
if [ "$lib" == "default" ]; then unset CPX_LIB_PATH; else export CPX_LIB_PATH=$PWD/$lib; fi
echo "$lib: $(timeout 120 python scripts/micro/ldpc_fixed_iters.py SPA 2>&1 | tail -2 | tr '\n' ' ')"
done
done
