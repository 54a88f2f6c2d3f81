#!/bin/bash
# gpurun_out/<tag>/* -> profiles/<tag>_*  (the judged copies; gpurun_out/ is scratch)     usage: bash scripts/copy_profiles.sh r04
TAG=${1:-r04}
R=$(cd $(dirname $0)/.. && pwd)
for f in $R/gpurun_out/$TAG/*; do
  b=$(basename $f)
  case $b in *.log) continue;; esac
  cp $f $R/profiles/${TAG}_$b
done
ls $R/profiles | grep "^${TAG}_" | wc -l
