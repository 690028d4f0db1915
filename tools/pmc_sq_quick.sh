#!/bin/bash
# Two --pmc passes (instruction counts; wave / wait cycles) over a short bench run: the per-graph instruction budget of the dominant kernel.
#   bash tools/pmc_sq_quick.sh <tag>   -> gpurun_out/<tag>_pmc3, _pmc4 (+ .log);  python profiles/summarize.py sq <db3> <db4> <out.json> --bench-log <log>
R=$GRAFT_REPO_ROOT; tag=$1; cd /tmp; export TMPDIR=/tmp
PMC_ARGS="--steps 6 --warmup 2 --cpu-sample 0 --no-secondary --skip-h2d-loop"
i=2
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $R/gpurun_out/${tag}_pmc$i -o p -- python3 $R/bench.py $PMC_ARGS > $R/gpurun_out/${tag}_pmc$i.log 2>&1 || exit 1
  echo "pmc pass $i done"
done
