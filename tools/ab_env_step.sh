#!/bin/bash
# A/B of a bench-side environment switch on the full step: alternating processes on one box, each tools/ab_knob.py FUSED 1
# (three blocks of 30 steps).     bash tools/ab_env_step.sh PN_BENCH_SAMPLE_BESIDE 0 1 [workload] [rounds]
VAR=$1; A=$2; B=$3; WL=${4:-cora}; N=${5:-3}
for i in $(seq $N); do
  for v in $A $B; do
    echo -n "$VAR=$v  "; env $VAR=$v python tools/ab_knob.py FUSED 1 workload=$WL blocks=3 steps=30 fused=1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().split('RESULT ')[1]); print(d['ms_per_step']['1'])"
  done
done
