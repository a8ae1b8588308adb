#!/bin/bash
# One GPU session that regenerates what profiles/ quotes for the current build (every step under its own timeout):
#   bash tools/final_pass.sh gpurun_out/final
set -u
OUT=${1:-gpurun_out/final}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --maxfail=10 > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
# exactly the driver's command; the stdout line must stay under 6 KB (round 5's 20 KB line came back unparsed)
timeout 420 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "stdout line bytes: $(tail -1 $OUT/bench.json | wc -c) (limit 6144), lines: $(wc -l < $OUT/bench.json)"
cp bench_extras.json $OUT/bench_extras.json
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/kt.log 2>&1
DB=$(find $OUT/kt -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB $OUT/kernel_stats.md > /dev/null
CSV=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
[ -n "$CSV" ] && cp $CSV $OUT/kernel_stats.csv
PMC_TIMEOUT=240 bash tools/pmc_passes.sh $OUT/pmc
# the traffic stamp of THIS build now exists: a second (short) bench line carries roofline.traffic / hbm_side
cp $OUT/pmc/pmc_traffic.json profiles/pmc_traffic.json
timeout 200 python bench.py --no-extras --no-cpu-baseline > $OUT/bench_traffic.json 2> $OUT/bench_traffic.err
PN_BENCH_FORCE_SHARDED=1 timeout 200 python bench.py --workload bgp --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $OUT/bgp_sharded_path.json 2> $OUT/bgp_sharded_path.err
tail -1 $OUT/bgp_sharded_path.err
# the N > 1 line as a bare command (bench.py starts its own ranks; two ranks share this one GPU: gloo, shrunk configs[3] / [4] blocks)
PN_BENCH_MULTI_SMALL=1 timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 > $OUT/bench_two_ranks_one_gpu.json 2> $OUT/bench_two_ranks_one_gpu.err
tail -1 $OUT/bench_two_ranks_one_gpu.err
ls $OUT $OUT/pmc
