#!/bin/bash
# A/B of the CUs the recurrent weight-gradient GEMM fills when the node-level GEMMs run beside it (profiles/r06_glue.txt sections
# 11-12): library variants built HERE first (no GPU needed),
#   cd pathnet_amd/csrc && mkdir -p _variants && OTHERS=$(ls _obj/*.o | grep -v pn_pagg.o) && for c in 200 216 224 232; do
#     hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DPN_WGRAD_CUS_SHARED=$c -c pn_pagg.hip -o _variants/p.o &&
#     hipcc -shared --offload-arch=gfx950 -o _variants/lib_cus$c.so $OTHERS _variants/p.o; done
# then on the GPU box alternating processes, fused step at the headline shape, medians of three 30-step blocks.
V=$GRAFT_REPO_ROOT/pathnet_amd/csrc/_variants
for i in 1 2; do
  for c in 208 200 216 224 232; do
    if [ $c = 208 ]; then L=""; else L="PN_LIB_PATH=$V/lib_cus$c.so"; fi
    echo -n "cus $c  "; env $L python tools/ab_knob.py FUSED 1 workload=cora blocks=3 steps=30 fused=1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().split('RESULT ')[1]); print(d['ms_per_step']['1'])"
  done
done
