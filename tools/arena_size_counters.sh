#!/bin/bash
# Address-translation and memory-side counters of the fused SVAT step at 10^6 columns for two arena sizes: 267 slots per tile (2.14 GB,
# the product) and 276 (2.21 GB, -DRH_STRIDE_PAD=9), DESIGN.md section 5.   gpurun -- tools/arena_size_counters.sh
# (build the variant first: python tools/build_variant.py pad9 --unit roger_hip -DRH_STRIDE_PAD=9)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out/arena_pmc
export RH_PMC_MODEL=svat RH_PMC_SIZE=1000x1000
pass=0
# (Round 2 ran a pass with TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum ... together; rocprofv3
# aborted and the call hung.  Cause, from that pass's own log: rocprofiler_create_counter_config failed with "error code 38:
# Request exceeds the capabilities of the hardware to collect" -- more TCP counters than the block has slots in ONE pass -- and the tool
# turns that into a fatal check (signal 6) at the first HIP call of the workload (inside rh_create, before any kernel had run): nothing
# to do with the fused kernel or its tail.  tools/arena_latency_counters.sh collects those counters ONE per pass.)
for counters in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"; do
  pass=$((pass + 1))
  for v in base pad9; do
    if [ $v = base ]; then unset ROGER_HIP_LIB; else export ROGER_HIP_LIB=roger_amd/variants/libroger_hip_$v.so; fi
    d=gpurun_out/arena_pmc/${v}_p$pass
    rm -rf $d
    rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $d -- python3 tools/pmc_workload.py > $d.log 2>&1 || tail -3 $d.log
    python3 - "$d" "$v" <<'PY'
import csv, glob, sys, collections
d, v = sys.argv[1], sys.argv[2]
vals, dur = collections.defaultdict(list), []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_step<2, false, true, false, false>" in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
print(f"{v:5s} kernel {sum(dur) / max(len(dur), 1) / 1e6:.4f} ms  " + "  ".join(f"{k} {sum(x) / len(x):.4g}" for k, x in sorted(vals.items())))
PY
  done
done
