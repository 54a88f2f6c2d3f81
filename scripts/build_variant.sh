#!/bin/bash
# A/B builds: bash scripts/build_variant.sh <file.hip> <out.so> -DMACRO ...   (links with the other objects of the in-tree build)
set -e
R=$(cd $(dirname $0)/.. && pwd)
SRC=$1; OUT=$2; shift 2
base=$(basename $SRC .hip)
mkdir -p /tmp/abbuild $(dirname $OUT)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden "$@" -I $R/include -I $R/commpy_amd/csrc -c $R/commpy_amd/csrc/$SRC -o /tmp/abbuild/${base}_variant.o
objs=$(ls $R/commpy_amd/csrc/build/*.o | grep -v "/${base}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $objs /tmp/abbuild/${base}_variant.o -ldl
echo built $OUT
