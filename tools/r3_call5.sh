#!/bin/bash
o=gpurun_out/r3; mkdir -p $o
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $o/pytest_gpu2.txt 2>&1; echo "pytest rc=$?"
tail -6 $o/pytest_gpu2.txt
timeout -k 10 300 python3 bench.py --model oned --routing --steps 200 --warmup 10 --no-cpu-baseline > $o/bench_routed.json 2> $o/bench_routed.err; echo "routed rc=$?"
RH_ROUTED_SEPARATE_GATHERS=1 timeout -k 10 300 python3 bench.py --model oned --routing --steps 200 --warmup 10 --no-cpu-baseline > $o/bench_routed_sep.json 2> $o/bench_routed_sep.err; echo "routed sep rc=$?"
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_driver.json 2> $o/bench_driver.err; echo "driver rc=$?"
python3 - <<'PY'
import json,glob
for f in ["gpurun_out/r3/bench_routed.json","gpurun_out/r3/bench_routed_sep.json","gpurun_out/r3/bench_driver.json"]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get("roofline",{})
        print(f, "value %.4g ms/step %.4f steps %d" % (d["value"], d["ms_per_step"], d["steps"]), "kernel %.4f frac %.3f" % (r["avg_kernel_ms"], r["frac"]))
    except Exception as e: print(f, "unreadable", e)
PY
