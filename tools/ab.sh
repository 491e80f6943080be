#!/bin/bash
# Same-box A/B of two builds of the library (VQHIP_LIB): tools/ab.sh <base.so> "<workloads>" [matched]
set -u
BASE=$1; shift
for WL in $1; do
  for rep in 1 2; do
    echo "== $WL base";  VQHIP_LIB=$BASE timeout -k 10 100 python -u tools/quick_bench.py $WL 10000000 ${2:-} 2>&1 | grep "samples/s" || exit 1
    echo "== $WL new";   timeout -k 10 100 python -u tools/quick_bench.py $WL 10000000 ${2:-} 2>&1 | grep "samples/s" || exit 1
  done
done
