#!/bin/bash
# round 5, GPU session F: dG as quads (PN_SEQH_DGQUAD) against the new default (saved values as quads) and round 4's layout (PN_SEQH_QUAD=0)
mkdir -p gpurun_out/r5f
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 600 python tools/tune_run.py 10 >> gpurun_out/r5f/tune_dgquad.txt 2>&1; done
cat gpurun_out/r5f/tune_dgquad.txt
PN_LIB_PATH=$GRAFT_REPO_ROOT/pathnet_amd/csrc/_variants/lib_1.so timeout 1200 python -m pytest tests/test_gpu_seqh.py tests/test_gpu_grad_error.py tests/test_gpu_pagg.py tests/test_gpu_batching.py tests/test_gpu_fused_step.py tests/test_gpu_determinism.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 | tee gpurun_out/r5f/pytest_dgquad.txt
