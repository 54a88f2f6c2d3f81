export TMPDIR=/tmp
for r in 1 2; do
for lib in default ab/v_ldpc_g1.so ab/v_ldpc_g2.so ab/v_ldpc_g3.so; do
if [ "$lib" == "default" ]; then unset CPX_LIB_PATH; else export CPX_LIB_PATH=$PWD/$lib; fi
echo "$lib: $(timeout 300 python scripts/micro/ldpc_fixed_iters.py 2>&1 | tail -3 | tr '\n' ' ')"
done
done
