#!/bin/bash
# VALU wave-instructions of the SAS kernel per column-day as a function of the sub-step count and of the age statistics: what one
# sub-step costs and what the rest of the day costs.   gpurun -- tools/sas_valu_split.sh [n_cells] [solver deterministic|Euler]
n=${1:-20000}
solver=${2:-deterministic}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out/valu_split
for cfg in "1 1" "2 1" "6 1" "6 0"; do
  set -- $cfg
  d=gpurun_out/valu_split/${solver}_s$1_st$2
  rm -rf $d
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $d -- python3 tools/sas_workload.py $n 3 $1 $2 $solver > $d.log 2>&1 || tail -3 $d.log
  python3 - "$d" "$n" "$1" "$2" <<'PY'
import csv, glob, sys, collections
d, n, sub, st = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
v = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_sas" in r["Kernel_Name"]:
            v[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "SQ_INSTS_VALU":
                v["ms"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6)
print(f"substeps {sub} stats {st}: " + "  ".join(f"{k} {sum(x) / len(x) / (1 if k == 'ms' else n):.1f}" for k, x in sorted(v.items())))
PY
done
