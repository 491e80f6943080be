#!/bin/bash
# Memory-path counters (TCP / TA / TD / TCC / SQ vmem) of one bench workload and of tools/hbm_stream beside it:
#   tools/pmc_mem.sh <tag> <workload>      -> gpurun_out/<tag>/mem.json  {kernel: {counter: mean per launch}}
set -u
TAG=${1:-mem}; WL=${2:-c2k8}
R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE"
B="TCC_BUSY_avr TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
timeout -s KILL 120 rocprofv3 --pmc $A --output-format csv -d $OUT/a_wl -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --cpu-rows 0 --no-extras --traffic off > $OUT/a_wl.log 2>&1 || { echo "pass A failed"; tail -5 $OUT/a_wl.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc $B --output-format csv -d $OUT/b_wl -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --cpu-rows 0 --no-extras --traffic off > $OUT/b_wl.log 2>&1 || { echo "pass B failed"; tail -5 $OUT/b_wl.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc $A --output-format csv -d $OUT/a_probe -- $R/tools/hbm_stream 10000000 1 > $OUT/a_probe.log 2>&1 || { echo "probe pass A failed"; tail -5 $OUT/a_probe.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc $B --output-format csv -d $OUT/b_probe -- $R/tools/hbm_stream 10000000 1 > $OUT/b_probe.log 2>&1 || { echo "probe pass B failed"; tail -5 $OUT/b_probe.log; exit 1; }
python3 - <<PY
import csv, glob, collections, json
out = {}
for d in ("a_wl", "b_wl", "a_probe", "b_probe"):
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % d):
        acc = collections.defaultdict(list); dur = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if not any(k in n for k in ("vq::k_", "k_grid", "k_block", "k_lds")) or "synth" in n or "crowd" in n:
                continue
            n = n.split("(")[0][:60]
            acc[(n, r["Counter_Name"])].append(float(r["Counter_Value"]))
            dur[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for (n, c), v in acc.items():
            out.setdefault(n, {})[c] = sum(v) / len(v)
        for n, v in dur.items():
            out.setdefault(n, {})["dispatch_ns_" + d[0]] = sum(v) / len(v)
json.dump(out, open("$OUT/mem.json", "w"), indent=1)
for n, c in out.items():
    print(n)
    for k, v in sorted(c.items()):
        print("    %-36s %.5g" % (k, v))
PY
