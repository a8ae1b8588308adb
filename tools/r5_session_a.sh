#!/bin/bash
# round 5, GPU session A: parity of the remainder-round tiling (forced sizes), then same-session A/B of the scatter / layout / tiling knobs
mkdir -p gpurun_out/r5a
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( PN_SEQH_TAIL=8 timeout 600 python -m pytest tests/test_gpu_seqh.py tests/test_gpu_pagg.py -x -q -m gpu 2>&1 | tail -5 ) > gpurun_out/r5a/pytest_tail8.txt
( PN_SEQH_TAIL=24 timeout 600 python -m pytest tests/test_gpu_seqh.py tests/test_gpu_pagg.py tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -5 ) > gpurun_out/r5a/pytest_tail24.txt
( timeout 600 python -m pytest tests/test_gpu_seqh.py tests/test_gpu_pagg.py tests/test_gpu_grad_error.py tests/test_gpu_batching.py tests/test_gpu_determinism.py tests/test_gpu_seq4.py tests/test_gpu_sampler.py -x -q -m gpu 2>&1 | tail -5 ) > gpurun_out/r5a/pytest_default.txt
( timeout 900 python tools/tune_run.py 10 2>&1 ) > gpurun_out/r5a/tune.txt
( timeout 900 python tools/tune_run.py 10 2>&1 ) > gpurun_out/r5a/tune_again.txt
tail -3 gpurun_out/r5a/pytest_*.txt; cat gpurun_out/r5a/tune.txt gpurun_out/r5a/tune_again.txt
