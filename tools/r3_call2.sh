#!/bin/bash
# round 3: full GPU suite + run() stepping modes
o=gpurun_out/r3; mkdir -p $o
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $o/pytest_gpu.txt 2>&1; echo "pytest rc=$?"
tail -8 $o/pytest_gpu.txt
for st in setup hooks; do
  timeout -k 10 300 python3 bench.py --stepping $st --days 20 --warmup-days 2 > $o/bench_svat_1e6_$st.json 2> $o/bench_svat_1e6_$st.err; echo "bench $st rc=$?"
done
timeout -k 10 300 python3 bench.py --steps 400 --warmup 40 --no-cpu-baseline > $o/bench_svat_1e6_400.json 2> $o/bench_svat_1e6_400.err; echo "bench rc=$?"
timeout -k 10 300 python3 bench.py --stepping setup --size 80 53 --days 60 --warmup-days 2 > $o/bench_svat_80x53_setup.json 2> $o/bench_svat_80x53_setup.err; echo "bench rc=$?"
timeout -k 10 300 python3 bench.py --stepping hooks --size 80 53 --days 60 --warmup-days 2 > $o/bench_svat_80x53_hooks.json 2> $o/bench_svat_80x53_hooks.err; echo "bench rc=$?"
timeout -k 10 300 python3 bench.py --size 80 53 --steps 2000 --warmup 100 --no-cpu-baseline > $o/bench_svat_80x53.json 2> $o/bench_svat_80x53.err; echo "bench rc=$?"
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get("roofline",{})
        print(f, "value %.4g ms/step %.4f steps %d" % (d["value"], d["ms_per_step"], d["steps"]), "kernel %.4f frac %.3f" % (r["avg_kernel_ms"], r["frac"]) if r else "", d["config"].get("stepping"))
    except Exception as e: print(f, "unreadable", e)
PY
