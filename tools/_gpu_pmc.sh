#!/bin/bash
# diagnostic: several --pmc passes over a short bench run (never combined with --stats); results under gpurun_out/pmc_<tag>_<i>/
# usage (on the GPU box, via gpurun):  bash tools/_gpu_pmc.sh <tag> "<counters pass 1>" "<counters pass 2>" ...
R=$GRAFT_REPO_ROOT; tag=$1; shift; cd /tmp; export TMPDIR=/tmp; i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $R/gpurun_out/pmc_${tag}_$i -o p -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-sample 0 > $R/gpurun_out/pmc_${tag}_$i.log 2>&1 || exit 1
done
