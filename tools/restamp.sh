set -u
OUT=gpurun_out/final7
mkdir -p $OUT
export TMPDIR=/tmp
timeout 100 python -m pytest tests/test_gpu_fused_step.py tests/test_gpu_optim.py tests/test_gpu_determinism.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
PMC_TIMEOUT=100 bash tools/pmc_passes.sh $OUT/pmc > /dev/null 2>&1
cat $OUT/pmc/summary.txt
cp $OUT/pmc/pmc_traffic.json profiles/pmc_traffic.json
timeout 150 python bench.py --no-extras --no-cpu-baseline > $OUT/bench_traffic.json 2> $OUT/bench_traffic.err
tail -c 400 $OUT/bench_traffic.json
