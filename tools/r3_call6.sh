#!/bin/bash
o=gpurun_out/r3; mkdir -p $o
timeout -k 10 900 python3 -m pytest tests/test_hip_routing.py tests/test_hip_sparse.py tests/test_hip_comm.py tests/test_bench_launcher.py -x -q -m gpu > $o/pytest_call6.txt 2>&1; echo "pytest rc=$?"
tail -6 $o/pytest_call6.txt
timeout -k 10 300 python3 bench.py --model oned --routing --steps 200 --warmup 10 --no-cpu-baseline > $o/bench_routed2.json 2> $o/bench_routed2.err; echo "routed rc=$?"
RH_NO_SPARSE_STORES=1 timeout -k 10 300 python3 bench.py --model oned --routing --steps 200 --warmup 10 --no-cpu-baseline > $o/bench_routed2_full.json 2> $o/bench_routed2_full.err; echo "routed full rc=$?"
python3 - <<'PY'
import json
for f in ["gpurun_out/r3/bench_routed2.json","gpurun_out/r3/bench_routed2_full.json"]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get("roofline",{})
        print(f, "value %.4g ms/step %.4f steps %d" % (d["value"], d["ms_per_step"], d["steps"]), "kernel %.4f frac %.3f" % (r["avg_kernel_ms"], r["frac"]), r["kernel"][:30], r["whole_step"])
    except Exception as e: print(f, "unreadable", e)
PY
tail -3 $o/bench_routed2.err
