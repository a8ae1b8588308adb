V=$GRAFT_REPO_ROOT/pathnet_amd/csrc/_variants
for i in 1 2; do
  for c in 208 200 216 224 232; do
    if [ $c = 208 ]; then L=""; else L="PN_LIB_PATH=$V/lib_cus$c.so"; fi
    echo -n "cus $c  "; env $L python tools/ab_knob.py FUSED 1 workload=cora blocks=3 steps=30 fused=1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().split('RESULT ')[1]); print(d['ms_per_step']['1'])"
  done
done
