#!/bin/bash
# SQ counter pass for the engine kernel: tools/pmc_sq.sh <tag> <workload> [VARIANT]
set -u
TAG=${1:-sq}; WL=${2:-c2}; export VQHIP_VARIANT=${3:-0}
R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq1 -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --cpu-rows 0 --no-extras --traffic off > $OUT/sq1.log 2>&1 || { echo "sq1 failed"; tail -5 $OUT/sq1.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_LEVEL_WAVES --output-format csv -d $OUT/sq2 -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --cpu-rows 0 --no-extras --traffic off > $OUT/sq2.log 2>&1 || { echo "sq2 failed"; tail -5 $OUT/sq2.log; exit 1; }
python3 - <<PY
import csv, glob, collections, json
out = {"workload": "$WL", "counters": {}}
for d in ("sq1", "sq2"):
    f = max(glob.glob("$OUT/%s/*/*counter_collection.csv" % d))
    acc = collections.defaultdict(list)
    recs = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in ("k_engine", "k_resident", "k_omega", "k_lowrank", "k_local", "k_direct"))]
    # the instantiation the launch spends its time in (the finish pass is another instantiation of the same template)
    spent = collections.Counter()
    for r in recs:
        spent[r["Kernel_Name"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    top = spent.most_common(1)[0][0] if spent else None
    for r in recs:
        if r["Kernel_Name"] == top:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            out["kernel"] = r["Kernel_Name"]
            out["last_dispatch_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, v in acc.items():
        out["counters"][k] = sum(v) / len(v)            # per launch, summed over the chip by rocprofv3
        print(f"{k:32s} {sum(v)/len(v):.6g}")
json.dump(out, open("$OUT/sq.json", "w"), indent=1)
PY
