#!/bin/bash
# SQ counters of a guard_bench case, per kernel instantiation (first pass and finish pass apart): tools/pmc_guard.sh <tag> <case>
set -u
TAG=${1:-pg}; CASE=${2:-clustered_256}
R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GUARD_STEPS=2
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq1 -- python3 $R/tools/guard_bench.py $CASE > $OUT/sq1.log 2>&1 || { echo "sq1 failed"; tail -5 $OUT/sq1.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/sq2 -- python3 $R/tools/guard_bench.py $CASE > $OUT/sq2.log 2>&1 || { echo "sq2 failed"; tail -5 $OUT/sq2.log; exit 1; }
python3 - <<PY
import csv, glob, collections, json
out = {}
for d in ("sq1", "sq2"):
    f = max(glob.glob("$OUT/%s/*/*counter_collection.csv" % d))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "vq::k_" in r["Kernel_Name"] and "synth" not in r["Kernel_Name"]:
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn, cs in acc.items():
        o = out.setdefault(kn, {})
        for k, v in cs.items():
            o[k] = sum(v) / len(v)
for kn, cs in out.items():
    print(kn)
    for k, v in cs.items():
        print(f"   {k:32s} {v:.6g}")
json.dump(out, open("$OUT/sq.json", "w"), indent=1)
PY
