#!/bin/bash
# Build variants of libpathnet_hip.so that differ in the -D knobs of pn_seq4.hip (HERE, no GPU needed):
#   bash tools/seq4_variants.sh "" "F4_NODMA=1" "F4_NOPHILOX=1 F4_NOSTORE=1" ...
# Each spec becomes pathnet_amd/csrc/_variants/lib_<n>.so (the other objects are the current build's); run them on the
# GPU box with tools/tune_run.py (PN_SEQ_MATH=bf16x3 and PN_SEQ4 in the environment select the kernels) or tools/trace_seq4.py.
set -e
cd "$(dirname "$0")/../pathnet_amd/csrc"
make -s -j4
mkdir -p _variants
rm -f _variants/lib_*.so
n=0
: > _variants/specs.txt
for spec in "$@"; do
  defs="-DPN_EXPERIMENTAL=1"     # the forward / BPTT of this file are outside the shipped library
  for kv in $spec; do defs="$defs -DPN_$kv"; done
  ( hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $defs -c pn_seq4.hip -o _variants/seq4_$n.o && \
    hipcc -shared --offload-arch=gfx950 -o _variants/lib_$n.so _obj/pn_host.o _obj/pn_sampler.o _obj/pn_pagg.o _obj/pn_train.o _obj/pn_merw.o _obj/pn_context.o _obj/pn_sort.o _obj/pn_rgrad.o _obj/pn_seqh.o _variants/seq4_$n.o && rm _variants/seq4_$n.o ) &
  echo "$n $spec" >> _variants/specs.txt
  n=$((n+1))
done
wait
cat _variants/specs.txt
