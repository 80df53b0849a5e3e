#!/bin/bash
o=gpurun_out/r3; mkdir -p $o
timeout -k 10 900 python3 -m pytest tests/test_hip_parity.py tests/test_restart_interchange.py tests/test_hip_host_package.py tests/test_hip_restart.py -x -q -m gpu -s -k "eberbaechle_full or restart or setup_step or stock_hooks or tutorial" > $o/pytest_call3.txt 2>&1; echo "pytest rc=$?"
grep -a "EBERBAECHLE\|passed\|failed" $o/pytest_call3.txt | tail -5
timeout -k 10 300 python3 tools/eberbaechle_full.py > $o/eberbaechle_full_80x53.json 2> $o/eberbaechle_full_80x53.err; echo "eb rc=$?"; cat $o/eberbaechle_full_80x53.json
for st in hooks; do
  timeout -k 10 300 python3 bench.py --stepping $st --days 20 --warmup-days 2 > $o/bench_svat_1e6_$st.json 2> $o/bench_svat_1e6_$st.err; echo "bench $st rc=$?"
  RH_STEP_CORE_UNSTAGED=1 timeout -k 10 300 python3 bench.py --stepping $st --days 20 --warmup-days 2 > $o/bench_svat_1e6_${st}_unstaged.json 2> $o/bench_svat_1e6_${st}_unstaged.err; echo "bench $st unstaged rc=$?"
done
timeout -k 10 300 python3 bench.py --stepping hooks --size 80 53 --days 60 --warmup-days 2 > $o/bench_svat_80x53_hooks.json 2> $o/bench_svat_80x53_hooks.err; echo "bench rc=$?"
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3/bench_*hooks*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value %.4g ms/step %.4f steps %d" % (d["value"], d["ms_per_step"], d["steps"]))
    except Exception as e: print(f, "unreadable", e)
PY
