#!/bin/bash
# Collect SQ / TA / TCP / TCC counter passes for the neighbour kernel (each pass its own run; no tracing options — gpurun refuses
# --pmc together with trace domains).  The library measured is $SPHMI_LIB (default: the in-tree build).
# usage: tools/pmc_passes.sh <outdir under gpurun_out/> [bench args]     (PMC_CMD='python … ' replaces the bench command, PMC_FILTER the
# kernel name the SQ / TA passes are summarised for, PMC_MAX=n stops after the first n passes)
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$out
echo "### library: ${SPHMI_LIB:-$R/sphexample_amd/libsphmi.so}"
python $R/tools/isa_report.py --lib ${SPHMI_LIB:-$R/sphexample_amd/libsphmi.so} --json $R/gpurun_out/$out/kernel_identity.json | grep -v '^    '
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  [ $i -gt ${PMC_MAX:-99} ] && break
  timeout 300 rocprofv3 --pmc $line -d $R/gpurun_out/$out/p$i -o p -- ${PMC_CMD:-python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --precondition-ms 0 "$@"} > $R/gpurun_out/$out/p$i.log 2>&1
  db=$(find $R/gpurun_out/$out/p$i -name '*.db' | head -1)
  echo "### pass $i: $line"
  case "$line" in
    FETCH_SIZE*|WRITE_SIZE*|TCC_*) python $R/tools/pmc_summary.py "$db" "sphmi::k_" ;;
    *) python $R/tools/pmc_summary.py "$db" "${PMC_FILTER:-k_neighbor_force}" ;;
  esac
  rm -rf $R/gpurun_out/$out/p$i
done <<'LIST'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_WAVE32_LDS
TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
GRBM_GUI_ACTIVE GRBM_COUNT
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum
LIST
