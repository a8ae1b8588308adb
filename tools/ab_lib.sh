#!/bin/bash
# A/B of two builds of the library on the full step, alternating processes on one box:
#   bash tools/ab_lib.sh <other .so> [workload] [rounds]      (the first column is the in-tree library)
OTHER=$1; WL=${2:-cora}; N=${3:-3}
for i in $(seq $N); do
  for l in "" "PN_LIB_PATH=$OTHER"; do
    echo -n "[${l:-in-tree}] "; env $l python tools/ab_knob.py FUSED 1 workload=$WL blocks=3 steps=30 fused=1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().split('RESULT ')[1]); print(d['ms_per_step']['1'])"
  done
done
