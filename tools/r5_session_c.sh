#!/bin/bash
# round 5, GPU session C: the test files session B had failures in (fixed since), then kernel traces of the eager step, the
# replayed hipGraph and the deterministic step (tools/graph_vs_eager.py)
mkdir -p gpurun_out/r5c
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( timeout 1200 python -m pytest tests/test_gpu_dist.py tests/test_merw_gen.py tests/test_gpu_batching.py tests/test_gpu_seqh.py tests/test_gpu_seq4.py tests/test_gpu_determinism.py -m gpu -q --tb=short -p no:cacheprovider --maxfail=20 2>&1 | tail -120 ) > gpurun_out/r5c/pytest_fixed.txt
tail -4 gpurun_out/r5c/pytest_fixed.txt
for mode in eager graph det; do
  timeout 300 rocprofv3 --kernel-trace -d gpurun_out/r5c/$mode -o t -- python tools/graph_vs_eager.py run $mode > gpurun_out/r5c/$mode.log 2>&1
  grep RESULT gpurun_out/r5c/$mode.log
done
python tools/graph_vs_eager.py analyse gpurun_out/r5c/eager gpurun_out/r5c/graph gpurun_out/r5c/det > gpurun_out/r5c/graph_vs_eager.txt 2>&1
grep RESULT gpurun_out/r5c/*.log >> gpurun_out/r5c/graph_vs_eager.txt
find gpurun_out/r5c -name "*.db" -size +40M -delete
cat gpurun_out/r5c/graph_vs_eager.txt | head -150
