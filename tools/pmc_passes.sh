#!/bin/bash
# rocprofv3 PMC passes over one short bench run (counters only: never combined with trace domains).
# usage (on the GPU box): bash tools/pmc_passes.sh <outdir> [kernel-regex]
set -u
OUT=${1:-gpurun_out/pmc}
RE=${2:-"seq_fwd|seq_bwd|wgrad3_kernel|gather_kernel|merw_walk"}
export TMPDIR=/tmp
mkdir -p $OUT
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
i=0
for SET in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
  "FETCH_SIZE TCC_HIT_sum" \
  "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" \
  "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" ; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-include-regex "$RE" --output-format csv -d $OUT/pass$i -o p$i -- $CMD > $OUT/pass$i.log 2>&1
  echo "pass $i rc=$?" >> $OUT/summary.txt
done
