#!/bin/bash
# One round of profiling evidence for the bench workload (run on the GPU box via gpurun; results under gpurun_out/<tag>_*):
#   1. rocprofv3 --kernel-trace --stats over a short bench run        -> per-kernel durations
#   2. separate --pmc passes (never combined with a trace domain)     -> FETCH_SIZE, WRITE_SIZE, SQ instruction / wait counters
# usage:  bash tools/profile_round.sh <tag>
R=$GRAFT_REPO_ROOT; tag=$1; cd /tmp; export TMPDIR=/tmp
# 4 plain passes make the batches resident (cold allocations: dropped by summarize.py --skip-first 4), then 5 warm-up + 20 timed steps
# of the resident loop: the average over those 25 launches IS the steady state bench.py's `value` is measured in
# --serial-steps: under rocprofv3 the runtime moves D2H copies with blit kernels (__amd_rocclr_copyBuffer) instead of the SDMA engines; overlapped
# with the decomposition kernel they take CUs from it (45.4 ms per launch against 42.7 ms alone and 42.9 ms in an unprofiled pipelined run)
ARGS="--steps 20 --warmup 5 --cpu-sample 0 --no-secondary --skip-h2d-loop --serial-steps"
PMC_ARGS="--steps 6 --warmup 2 --cpu-sample 0 --no-secondary --skip-h2d-loop"      # counters are per launch: no need for a long run
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_stats -o s -- python3 $R/bench.py $ARGS > $R/gpurun_out/${tag}_stats.log 2>&1 || exit 1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $R/gpurun_out/${tag}_pmc$i -o p -- python3 $R/bench.py $PMC_ARGS > $R/gpurun_out/${tag}_pmc$i.log 2>&1 || exit 1
  echo "pmc pass $i done"
done
