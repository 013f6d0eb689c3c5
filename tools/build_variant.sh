#!/bin/bash
# Builds a tuning variant of the library: one source recompiled with extra flags, linked with the
# other objects of the regular build (run `make -C n-body-simulation_amd` first) into
# n-body-simulation_amd/variants/libnbody_<name>.so; NBODY_HIP_LIB=<that file> selects it.
# usage: tools/build_variant.sh <name> <source stem, e.g. nb_bh> "<flags>" [<name> <stem> "<flags>" ...]
set -e
cd "$(dirname "$0")/../n-body-simulation_amd"
mkdir -p variants build/var
COMMON="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -I../include -Icsrc"
while [ $# -gt 2 ]; do
  name=$1; stem=$2; flags=$3; shift 3
  /opt/rocm/bin/hipcc $COMMON $flags -c csrc/$stem.hip -o build/var/${stem}_$name.o
  objs=""
  for o in build/*.o; do
    if [ "$o" = "build/$stem.hip.o" ]; then objs="$objs build/var/${stem}_$name.o"; else objs="$objs $o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libnbody_$name.so $objs
  echo built $name
done
