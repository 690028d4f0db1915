#!/bin/bash
# Issue / latency counters of the decomposition kernel over short bench runs (diagnostic; run on the GPU box via gpurun):
#   bash tools/pmc_sq_detail.sh <tag>      -> gpurun_out/<tag>_sq<k>/   (summaries: python profiles/summarize.py sq <dbs...> <out.json>)
R=$GRAFT_REPO_ROOT; tag=$1; cd /tmp; export TMPDIR=/tmp
PMC_ARGS="--steps 6 --warmup 2 --cpu-sample 0 --no-secondary --skip-h2d-loop"
i=0
for set in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
           "SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" \
           "SQ_INSTS_VALU_CVT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $R/gpurun_out/${tag}_sq$i -o p -- python3 $R/bench.py $PMC_ARGS > $R/gpurun_out/${tag}_sq$i.log 2>&1 || { echo "pmc set $i failed"; tail -3 $R/gpurun_out/${tag}_sq$i.log; }
  echo "sq detail pass $i done"
done
