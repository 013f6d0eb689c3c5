#!/bin/bash
# Builds tuning variants of the library (different NB_DS_* settings) into variants/.
# usage: build_variants.sh name "flags" [name "flags" ...]
set -e
cd "$(dirname "$0")"
mkdir -p variants build/var
COMMON="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -I../include -Icsrc"
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc $COMMON $flags -c csrc/nb_direct.hip -o build/var/nb_direct_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libnbody_$name.so build/nb_api.hip.o build/var/nb_direct_$name.o build/nb_bh.hip.o build/nb_diag.hip.o build/nb_static.hip.o build/nb_comm.hip.o build/nb_refresh.hip.o build/nb_api3.hip.o build/nb_direct3.hip.o build/nb_bh3.hip.o build/nb_scenario.cpp.o
  echo built $name
done
