#!/bin/bash
# round 6: rocprofv3 kernel traces of the headline step in its three shapes -- three library calls, pn_pagg_train_step with the pooling
# step as one launch reading global memory (PN_POOL_STEP=2) and with its tiles staged in LDS (=1) -- cut into steps by tools/graph_vs_eager.py
OUT=${1:-gpurun_out/r6trace}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PN_BENCH_FUSED=0 timeout 300 rocprofv3 --kernel-trace -d $OUT/three -o t -- python tools/graph_vs_eager.py run eager > $OUT/three.log 2>&1
PN_BENCH_FUSED=1 PN_POOL_STEP=2 timeout 300 rocprofv3 --kernel-trace -d $OUT/fused_global -o t -- python tools/graph_vs_eager.py run eager > $OUT/fused_global.log 2>&1
PN_BENCH_FUSED=1 PN_POOL_STEP=1 timeout 300 rocprofv3 --kernel-trace -d $OUT/fused_staged -o t -- python tools/graph_vs_eager.py run eager > $OUT/fused_staged.log 2>&1
python tools/graph_vs_eager.py analyse $OUT/three $OUT/fused_global $OUT/fused_staged > $OUT/trace.txt 2>&1
find $OUT -name "*.db" -delete
cat $OUT/trace.txt
