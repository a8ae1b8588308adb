#!/bin/bash
# tuning: the CUs the weight-gradient kernel may fill (PN_WGRAD_CUS) -- its 8-wave workgroups take a CU's whole register
# file, so with all 256 the node-level GEMMs of the main stream cannot start until it is done.   (GPU box)
mkdir -p gpurun_out/ab5
for wl in cora pubmed bgp; do
  case $wl in cora) st=200;; pubmed) st=40;; bgp) st=15;; esac
  for i in 1 2; do for c in ${CUS:-256 240 232 224 216}; do
    PN_WGRAD_CUS=$c python bench.py --workload $wl --steps $st --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms']; print('$wl cus $c step', round(d['ms_per_step'],4), 'wgrad', s['wgrad'])"
  done; done
done 2>&1 | tee gpurun_out/ab5/wgrad_cus.txt
