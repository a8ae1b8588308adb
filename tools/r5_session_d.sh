#!/bin/bash
# round 5, GPU session D: (1) CUs the recurrent weight-gradient launch may fill, below round 4's 216 (the kernel trace of session C shows
# the node-level weight-gradient GEMMs, squeezed onto 32 CUs beside it, ending the step 38 us after it); (2) the deterministic step
# after the four-deep scatter walk
mkdir -p gpurun_out/r5d
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for round in 0 1; do
for cus in 224 208 192 176 160 128; do
  PN_WGRAD_CUS=$cus timeout 200 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline --no-graph 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
b = d['dispersion']['block_ms_per_step']
print('cora cus $cus step %.4f median-of-5 %.4f min %.4f wgrad %.4f bank_bwd %.4f fc0_bwd %.4f' % (d['ms_per_step'], b['median'], b['min'], d['stages_ms']['wgrad'], d['stages_ms']['bank_bwd'], d['stages_ms']['fc0_bwd']))
" >> gpurun_out/r5d/wgrad_cus.txt
done
done
cat gpurun_out/r5d/wgrad_cus.txt
for mode in eager det eager det; do python tools/graph_vs_eager.py run $mode 2>/dev/null | grep RESULT >> gpurun_out/r5d/det.txt; done
cat gpurun_out/r5d/det.txt
