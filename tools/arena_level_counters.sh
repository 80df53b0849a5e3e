#!/bin/bash
# Address-translation counters of the fused step on a fast and on a slow arena of the same process (tools/arena_levels.py shows the
# levels).    gpurun -- tools/arena_level_counters.sh
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
d=gpurun_out/arena_level_pmc
rm -rf $d; mkdir -p $d
rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $d -- python3 tools/arena_level_workload.py > $d.log 2>&1 || tail -3 $d.log
python3 - "$d" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_step<2, false, true, false, false>" in r["Kernel_Name"]:
            rows.append(r)
by = collections.defaultdict(dict)
for r in rows:
    key = int(r["Dispatch_Id"])
    by[key][r["Counter_Name"]] = float(r["Counter_Value"])
    by[key]["dur"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
ids = sorted(by)
tail = ids[-58:]          # the 2 x 29 lazy launches of the measured runs (each run starts with one launch that may be eager)
for name, sel in (("first arena", tail[:27]), ("last arena ", tail[-27:])):
    keys = sorted(k for k in by[sel[0]] if k != "dur")
    print(name, f"kernel {sum(by[i]['dur'] for i in sel) / len(sel) / 1e6:.4f} ms  " + "  ".join(f"{k} {sum(by[i][k] for i in sel) / len(sel):.5g}" for k in keys))
PY
