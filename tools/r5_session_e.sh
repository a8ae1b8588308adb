#!/bin/bash
# round 5, GPU session E: the BPTT's saved values as one 16-byte quad per (path step, unit) -- 16 dwordx4 loads per step instead of 64 dword loads
mkdir -p gpurun_out/r5e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 600 python tools/tune_run.py 10 >> gpurun_out/r5e/tune_quad.txt 2>&1; done
cat gpurun_out/r5e/tune_quad.txt
PN_LIB_PATH=$GRAFT_REPO_ROOT/pathnet_amd/csrc/_variants/lib_1.so timeout 900 python -m pytest tests/test_gpu_seqh.py tests/test_gpu_grad_error.py tests/test_gpu_pagg.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 | tee gpurun_out/r5e/pytest_quad.txt
