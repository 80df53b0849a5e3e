#!/bin/bash
# A few SQ counters of the SAS kernels, deterministic against Euler, per column-day.   gpurun -- tools/sas_pmc_compare.sh [n_cells]
n=${1:-20000}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out/sas_pmc
pass=0
for counters in "SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SALU" \
                "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
                "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES"; do
  pass=$((pass + 1))
  for solver in deterministic Euler; do
    d=gpurun_out/sas_pmc/${solver}_p$pass
    rm -rf $d
    rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $d -- python3 tools/sas_workload.py $n 3 6 1 $solver > $d.log 2>&1 || tail -3 $d.log
    python3 - "$d" "$n" "$solver" <<'PY'
import csv, glob, sys, collections
d, n, solver = sys.argv[1], int(sys.argv[2]), sys.argv[3]
v = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_sas" in r["Kernel_Name"]:
            v[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"{solver:14s} " + "  ".join(f"{k} {sum(x) / len(x) / n:.1f}" for k, x in sorted(v.items())))
PY
  done
done
