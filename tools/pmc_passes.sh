#!/bin/bash
# Collect SQ/TA/TCP counter passes for the force kernel (each pass its own run; no tracing options).
# usage: tools/pmc_passes.sh <outdir> [bench args]     (PMC_CMD='python … ' replaces the bench command)
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$out
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $line -d $R/gpurun_out/$out/p$i -o p -- ${PMC_CMD:-python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --precondition-ms 0 "$@"} > $R/gpurun_out/$out/p$i.log 2>&1
  db=$(find $R/gpurun_out/$out/p$i -name '*.db' | head -1)
  echo "### pass $i: $line"
  python $R/tools/pmc_summary.py "$db" "k_neighbor_force" | tail -n +2
done <<'LIST'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_WAVE32_LDS
TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
GRBM_GUI_ACTIVE GRBM_COUNT
LIST
