#!/bin/bash
# Runs on the GPU box (via gpurun): the round's profiles for every bench workload.
# usage: tools/profile_round.sh <round-tag> "<workloads>"     outputs under gpurun_out/<tag>_<workload>/
set -u
TAG=${1:-r02}; shift
for WL in ${1:-c2 c3 c4 c5 c2k32 c2k8}; do
    bash tools/profile_gpu.sh ${TAG}_${WL} $WL > gpurun_out/${TAG}_${WL}.txt 2>&1 || { echo "profile of $WL failed"; tail -5 gpurun_out/${TAG}_${WL}.txt; exit 1; }
    echo "profiled $WL"
done
