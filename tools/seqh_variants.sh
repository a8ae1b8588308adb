#!/bin/bash
# Build variants of libpathnet_hip.so that differ in the knobs of pn_seqh.hip (HERE, no GPU needed):
#   bash tools/seqh_variants.sh "" "FWDH_WAVES=2" "TRACE_H=1" "env:PN_SEQH_TAIL=0" ...
# Each spec becomes pathnet_amd/csrc/_variants/lib_<n>.so (run: tools/tune_run.py / tools/trace_seqh.py with PN_LIB_PATH).
set -e
cd "$(dirname "$0")/../pathnet_amd/csrc"
make -s -j4 > /dev/null
mkdir -p _variants
n=0
: > _variants/specs.txt
OTHERS=$(ls _obj/*.o | grep -v pn_seqh.o)
for spec in "$@"; do
  defs=""
  for kv in $spec; do case $kv in env:*) ;; *) defs="$defs -DPN_$kv";; esac; done     # (env:NAME=VALUE tokens: run-time knobs, tools/tune_run.py)
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $defs -c pn_seqh.hip -o _variants/seqh_$n.o && \
    /opt/rocm/bin/hipcc -shared --offload-arch=gfx950 -o _variants/lib_$n.so $OTHERS _variants/seqh_$n.o && rm _variants/seqh_$n.o ) &
  echo "$n $spec" >> _variants/specs.txt
  n=$((n+1))
done
wait
cat _variants/specs.txt
