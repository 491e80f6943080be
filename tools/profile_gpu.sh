#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + the two HBM PMC passes for one bench workload.
# usage: tools/profile_gpu.sh <tag> [workload]      outputs under gpurun_out/<tag>/
set -u
TAG=${1:-prof}; WL=${2:-c2}
R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --workload $WL --steps 40 --warmup 3 --cpu-rows 0 --no-extras --traffic off > $OUT/kt.log 2>&1 || { echo "kernel-trace failed"; tail -5 $OUT/kt.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --cpu-rows 0 --no-extras --traffic off > $OUT/pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; tail -5 $OUT/pmc_fetch.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --cpu-rows 0 --no-extras --traffic off > $OUT/pmc_write.log 2>&1 || { echo "pmc write failed"; tail -5 $OUT/pmc_write.log; exit 1; }
find $OUT -name "*.csv" | head -20
