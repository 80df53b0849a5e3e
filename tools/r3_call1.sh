#!/bin/bash
# round 3, call: sparse-store tests + bench + chunk experiment
o=gpurun_out/r3; mkdir -p $o
timeout -k 10 200 roger_amd/variants/chunk_levels 16 1088 > $o/chunk_levels.txt 2>&1 || echo "chunk_levels failed"
timeout -k 10 900 python3 -m pytest tests/test_hip_sparse.py tests/test_hip_parity.py -x -q -m gpu -k "sparse or pure_output or pieces or accumulated or hooks_phase or lazy or device_hooks or tail_control" > $o/pytest_sparse.txt 2>&1; echo "pytest rc=$?"
tail -5 $o/pytest_sparse.txt
timeout -k 10 300 python3 bench.py --steps 60 --warmup 5 > $o/bench_svat_1e6.json 2> $o/bench_svat_1e6.err; echo "bench rc=$?"
RH_NO_SPARSE_STORES=1 timeout -k 10 300 python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline > $o/bench_svat_1e6_nosparse.json 2> $o/bench_svat_1e6_nosparse.err; echo "bench rc=$?"
timeout -k 10 300 python3 bench.py --steps 60 --warmup 5 --model oned --no-cpu-baseline > $o/bench_oned_1e6.json 2> $o/bench_oned_1e6.err; echo "bench rc=$?"
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print(f, "value %.4g ms/step %.4f kernel %.4f frac %.3f" % (d["value"], d["ms_per_step"], r["avg_kernel_ms"], r["frac"]), r["kernel"], r.get("full_store_launch"), d["config"].get("placement_probe_ms"))
    except Exception as e: print(f, "unreadable", e)
PY
