#!/bin/bash
# round 5, GPU session G: the step with the next epoch's paths prefetched on a sampling stream vs strictly in sequence
mkdir -p gpurun_out/r5g
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_bench_step.py tests/test_gpu_optim.py -m gpu -q -p no:cacheprovider 2>&1 | tail -5 | tee gpurun_out/r5g/pytest.txt
for round in 0 1 2; do
for pf in 0 1; do
  PN_BENCH_PREFETCH=$pf timeout 200 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline --no-graph 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
b = d['dispersion']['block_ms_per_step']
print('prefetch $pf step %.4f median-of-5 %.4f min %.4f  roofline %s %.4f' % (d['ms_per_step'], b['median'], b['min'], d['roofline']['bound'], d['roofline']['frac']))
" >> gpurun_out/r5g/prefetch_ab.txt
done
done
cat gpurun_out/r5g/prefetch_ab.txt
