#!/bin/bash
# SQ counters of the fused SVAT step (what bounds it: VALU issue, memory, waiting?), per launch of each k_step variant.
#   gpurun -- tools/svat_pmc.sh [model=svat] [size=1000x1000]
export RH_PMC_MODEL=${1:-svat} RH_PMC_SIZE=${2:-1000x1000}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
o=gpurun_out/svat_pmc; mkdir -p $o
pass=0
for counters in "SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
                "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
                "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64" \
                "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  pass=$((pass + 1))
  d=$o/p$pass; rm -rf $d
  rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $d -- python3 tools/pmc_workload.py > $d.log 2>&1 || tail -3 $d.log
done
python3 - "$o" <<'PY'
import csv, glob, sys, collections
o = sys.argv[1]
v = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(o + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if k.startswith("void k_step<") or "k_calib" in k:
            k = k.split("(")[0]
            v[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k in sorted(v):
    n = max(len(x) for x in v[k].values())
    print(f"== {k}: ~{n} launches, {sum(dur[k]) / len(dur[k]) / 1e3:.1f} us under the counters")
    for c, x in sorted(v[k].items()):
        print(f"   {c:28s} {sum(x) / len(x):.4g}")
PY
