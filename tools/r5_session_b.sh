#!/bin/bash
# round 5, GPU session B: the whole GPU suite on the new build (relative gradient checks logged), the N = 1 bench line, the
# self-spawned two-rank bench
mkdir -p gpurun_out/r5b
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PN_GRADCHECK_LOG=$GRAFT_REPO_ROOT/gpurun_out/r5b/gradcheck.jsonl
( timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --maxfail=20 2>&1 | tail -60 ) > gpurun_out/r5b/pytest_gpu.txt
unset PN_GRADCHECK_LOG
tail -5 gpurun_out/r5b/pytest_gpu.txt
( timeout 600 python bench.py > gpurun_out/r5b/bench.json 2> gpurun_out/r5b/bench.err ); tail -2 gpurun_out/r5b/bench.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r5b/bench.json"))
print("ms_per_step", d["ms_per_step"], "dispersion", d["dispersion"]["block_ms_per_step"], d["dispersion"]["step_ms"])
print("roofline", {k: d["roofline"][k] for k in ("kernel","bound","achieved","peak","unit","frac")}, d["roofline"]["bound_evidence"])
print("stages", d["stages_ms"])
print("det", d.get("headline_step_deterministic"))
print("graph", d.get("graph_replay"))
print("pubmed", d["pubmed_scale_step"]["ms_per_step"], d["pubmed_scale_step"]["deterministic"]["deterministic_ms_per_step"])
print("bgp", d["bgp_scale_step"]["ms_per_step"], "c4", d["configs4_one_gpu_step"].get("seconds_per_step"))
print("glibc", d.get("sampler_glibc_replay"), d.get("sampler", {}).get("value"))
P
