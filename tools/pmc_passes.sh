#!/bin/bash
# rocprofv3 PMC passes over short bench runs (counters only: never combined with trace domains).
# usage (on the GPU box): bash tools/pmc_passes.sh <outdir>
# Writes <outdir>/{cora,pubmed}/pass*/..., <outdir>/pmc_summary_{cora,pubmed}.md and <outdir>/pmc_traffic.json
# (stamped with the library's source hash: bench.py quotes it only for the same build).
set -u
OUT=${1:-gpurun_out/pmc}
RE="seq_fwdh|seq_bwdh|wgradh_kernel|seq_fwd3|seq_bwd3|wgrad3_kernel|wgrad4_kernel|gather_kernel|merw_walk"
export TMPDIR=/tmp
mkdir -p $OUT
run_set() {   # workload, pass number, counters...
  local WL=$1; local N=$2; shift 2
  mkdir -p $OUT/$WL
  timeout ${PMC_TIMEOUT:-240} rocprofv3 --pmc "$@" --kernel-include-regex "$RE" --output-format csv -d $OUT/$WL/pass$N -o p$N -- \
     python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --workload $WL > $OUT/$WL/pass$N.log 2>&1
  echo "$WL pass $N rc=$?" >> $OUT/summary.txt
}
run_set cora 1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU
run_set cora 2 FETCH_SIZE TCC_HIT_sum
run_set cora 3 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
if [ "${PMC_PUBMED:-1}" = "1" ]; then
  run_set pubmed 2 FETCH_SIZE TCC_HIT_sum
  run_set pubmed 3 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
fi
python tools/pmc_summary.py $OUT/cora $OUT/pmc_summary_cora.md > /dev/null
[ -d $OUT/pubmed ] && python tools/pmc_summary.py $OUT/pubmed $OUT/pmc_summary_pubmed.md > /dev/null
python tools/pmc_traffic.py $OUT $OUT/pmc_traffic.json
