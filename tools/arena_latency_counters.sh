#!/bin/bash
# Memory-side latency / queue counters of the fused step on the first and on the last arena of one process (the arenas sit at different
# speed levels, tools/arena_levels.py), ONE counter per pass: several TCP latency counters in one pass exceed the block's slots and
# rocprofv3 aborts (tools/arena_size_counters.sh).     gpurun -- tools/arena_latency_counters.sh [contexts=6]
export RH_LEVEL_CONTEXTS=${1:-6}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
o=gpurun_out/arena_latency; rm -rf $o; mkdir -p $o
for c in TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum \
         TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WR_UNCACHED_32B_sum; do
  rocprofv3 --pmc $c GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $o/$c -- python3 tools/arena_level_workload.py > $o/$c.log 2>&1 || { echo "pass $c failed: $(grep -m1 -i 'error code\|fatal' $o/$c.log)"; continue; }
done
python3 - "$o" <<'PY'
import csv, glob, sys, collections
o = sys.argv[1]
for d in sorted(glob.glob(o + "/*/")):
    by = collections.defaultdict(dict)
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_step<2, false, true, true, false>" in r["Kernel_Name"]:
                k = int(r["Dispatch_Id"])
                by[k][r["Counter_Name"]] = float(r["Counter_Value"])
                by[k]["dur"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    ids = sorted(by)
    if len(ids) < 58:
        print(d, "too few launches", len(ids)); continue
    tail = ids[-58:]
    for name, sel in (("first arena", tail[:27]), ("last arena ", tail[-27:])):
        keys = sorted(k for k in by[sel[0]] if k != "dur")
        print(f"{d.split('/')[-2]:32s} {name} kernel {sum(by[i]['dur'] for i in sel) / len(sel) / 1e3:7.1f} us  " + "  ".join(f"{k} {sum(by[i][k] for i in sel) / len(sel):.5g}" for k in keys))
PY
