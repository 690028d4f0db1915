#!/bin/bash
# Instruction-fetch counters of the decomposition kernel over a short bench run (diagnostic; run on the GPU box via gpurun):
#   bash tools/pmc_icache.sh <tag>      -> gpurun_out/<tag>_avail.txt, gpurun_out/<tag>_ic<k>/
R=$GRAFT_REPO_ROOT; tag=$1; cd /tmp; export TMPDIR=/tmp
PMC_ARGS="--steps 6 --warmup 2 --cpu-sample 0 --no-secondary --skip-h2d-loop"
rocprofv3 --list-avail > $R/gpurun_out/${tag}_avail.txt 2>&1
i=0
for set in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $R/gpurun_out/${tag}_ic$i -o p -- python3 $R/bench.py $PMC_ARGS > $R/gpurun_out/${tag}_ic$i.log 2>&1 || { echo "pmc set $i failed"; tail -3 $R/gpurun_out/${tag}_ic$i.log; }
  echo "icache pass $i done"
done
