#!/bin/bash
# Instruction-cache behaviour of the SAS kernel (539 KB of code, 64 KB of instruction cache per two CUs): requests, hits, misses.
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
o=gpurun_out/sas_icache; rm -rf $o; mkdir -p $o
rocprofv3 -L > $o/counters_list.txt 2>&1
grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQC_TC_INST[A-Z_]*\|SQ_WAIT_INST[A-Z_]*" $o/counters_list.txt | sort -u | tr '\n' ' '; echo
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $o/p1 -- python3 tools/sas_workload.py 100000 3 > $o/p1.log 2>&1 || tail -3 $o/p1.log
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $o/p2 -- python3 tools/sas_workload.py 100000 3 > $o/p2.log 2>&1 || tail -3 $o/p2.log
python3 - <<'PY'
import csv, glob, collections
for p in ("p1", "p2"):
    v = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/sas_icache/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_sas" in r["Kernel_Name"]:
                v[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(p, {k: f"{sum(x)/len(x):.4g}" for k, x in v.items()})
PY
