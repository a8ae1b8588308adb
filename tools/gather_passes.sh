#!/bin/bash
# rocprofv3 evidence for the gather figures bench.py prints: kernel trace, then the FETCH_SIZE / WRITE_SIZE passes
# (counters only, never combined with a trace domain).      bash tools/gather_passes.sh gpurun_out/gather
OUT=${1:-gpurun_out/gather}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
RE="gather_kernel|seq_fwdh_kernel"
python tools/gather_probe.py 10 2>&1 | grep RESULT > $OUT/probe.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python tools/gather_probe.py 6 > $OUT/kt.log 2>&1
CSV=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); [ -n "$CSV" ] && grep -E "Name|$RE" $CSV > $OUT/kernel_stats.csv
timeout 300 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --kernel-include-regex "$RE" --output-format csv -d $OUT/pass2 -o p2 -- python tools/gather_probe.py 4 > $OUT/pass2.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum --kernel-include-regex "$RE" --output-format csv -d $OUT/pass3 -o p3 -- python tools/gather_probe.py 4 > $OUT/pass3.log 2>&1
python - <<PY > $OUT/summary.md
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0].strip()
        acc[n][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("| kernel | launches | FETCH_SIZE KB | WRITE_SIZE KB | HBM bytes = (2 FETCH + WRITE) KiB | TCC_HIT | TCC_MISS | TCC_REQ |")
print("|---|---|---|---|---|---|---|---|")
for k, cs in acc.items():
    m = lambda c: sum(cs[c]) / len(cs[c]) if cs.get(c) else float("nan")
    print("| %s | %d | %.0f | %.0f | %.3e | %.3e | %.3e | %.3e |" % (k, len(cs.get("FETCH_SIZE", [])), m("FETCH_SIZE"), m("WRITE_SIZE"),
          (2 * m("FETCH_SIZE") + m("WRITE_SIZE")) * 1024, m("TCC_HIT_sum"), m("TCC_MISS_sum"), m("TCC_REQ_sum")))
PY
find $OUT -name "*.db" -delete
cat $OUT/probe.txt $OUT/kernel_stats.csv $OUT/summary.md
