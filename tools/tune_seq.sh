#!/bin/bash
# Build variants of libpathnet_hip.so with different recurrent-kernel knobs (HERE, no GPU needed):
#   bash tools/tune_seq.sh "FWD_MT=64 FWD_WAVES=1" "FWD_MT=32 FWD_DEPTH=2" ...
# Each spec becomes pathnet_amd/csrc/_variants/lib_<n>.so; run on the GPU box with tools/tune_run.py.
set -e
cd "$(dirname "$0")/../pathnet_amd/csrc"
mkdir -p _variants _obj
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -x hip -c pn_host.cpp -o _obj/pn_host.o
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c pn_sampler.hip -o _obj/pn_sampler.o
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c pn_train.hip -o _obj/pn_train.o
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c pn_merw.hip -o _obj/pn_merw.o
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c pn_context.hip -o _obj/pn_context.o
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c pn_seq4.hip -o _obj/pn_seq4.o
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c pn_sort.hip -o _obj/pn_sort.o
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c pn_rgrad.hip -o _obj/pn_rgrad.o
n=0
: > _variants/specs.txt
for spec in "$@"; do
  defs=""
  for kv in $spec; do defs="$defs -DPN_$kv"; done
  ( hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $defs -c pn_pagg.hip -o _variants/pagg_$n.o && \
    hipcc -shared --offload-arch=gfx950 -o _variants/lib_$n.so _obj/pn_host.o _obj/pn_sampler.o _obj/pn_train.o _obj/pn_merw.o _obj/pn_context.o _obj/pn_seq4.o _obj/pn_sort.o _obj/pn_rgrad.o _variants/pagg_$n.o && rm _variants/pagg_$n.o ) &
  echo "$n $spec" >> _variants/specs.txt
  n=$((n+1))
done
wait
cat _variants/specs.txt
