#!/bin/bash
# From the build container (scripts/install_hooks.sh keeps .git_head current per commit; this restamps on EVERY call): stamp the commit into .git_head (the GPU box gets the tree without .git), then run
# scripts/gpu_round.sh there in ONE gpurun call.   usage: bash scripts/run_gpu_round.sh [tag] [sections] [timeout_s]
R=$(cd $(dirname $0)/.. && pwd)
cd $R
head=$(git rev-parse HEAD)
if ! git diff --quiet HEAD -- . ':!profiles' ':!*.md'; then head="$head+dirty"; fi
echo "$head" > .git_head
TAG=${1:-r05}; SEC=${2:-tsbdklvumxcfLTohRP}; TMO=${3:-1500}
exec /usr/local/graft/bin/gpurun --timeout $TMO -- "bash scripts/gpu_round.sh $TAG $SEC"
