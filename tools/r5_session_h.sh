#!/bin/bash
# round 5, GPU session H: deterministic mode with ONE sort for the three scatters' orders and the pooling backward's two scatters in one launch per
# pass; three ranks on one GPU with the product kernels
mkdir -p gpurun_out/r5h
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_fused_step.py tests/test_gpu_dist.py -m gpu -q --tb=short -p no:cacheprovider --maxfail=10 2>&1 | tail -30 > gpurun_out/r5h/pytest.txt
tail -4 gpurun_out/r5h/pytest.txt
for mode in eager det eager det; do python tools/graph_vs_eager.py run $mode 2>/dev/null | grep RESULT >> gpurun_out/r5h/det.txt; done
cat gpurun_out/r5h/det.txt
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/r5h/det -o t -- python tools/graph_vs_eager.py run det > gpurun_out/r5h/det.log 2>&1
python tools/graph_vs_eager.py analyse gpurun_out/r5h/det > gpurun_out/r5h/det_trace.txt 2>&1
find gpurun_out/r5h -name "*.db" -delete
head -60 gpurun_out/r5h/det_trace.txt
