#!/bin/bash
# Runs tools/traverse_ab.py for the default library (tag "default") and for every variant tag given
# (n-body-simulation_amd/variants/libnbody_<tag>.so), one process each, then compares each build's
# accelerations of the initial state with the first tag's.  usage: tools/run_ab.sh <late step> <n> <tag>...
set -e
late=$1; n=$2; shift 2
V=n-body-simulation_amd/variants
first=$1
for t in "$@"; do
  if [ "$t" = default ]; then
    timeout -k 10 200 python3 tools/traverse_ab.py $late $n default
  else
    NBODY_HIP_LIB=$V/libnbody_$t.so timeout -k 10 200 python3 tools/traverse_ab.py $late $n $t
  fi
done
for t in "$@"; do
  [ "$t" = "$first" ] || python3 tools/traverse_ab.py diff $first $t
done
