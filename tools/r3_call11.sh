#!/bin/bash
o=gpurun_out/r3; mkdir -p $o
timeout -k 10 900 python3 -m pytest tests/test_hip_sas.py tests/test_hip_sas_bromide.py tests/test_hip_host_sas.py -x -q -m gpu > $o/pytest_sas_solv.txt 2>&1; echo "pytest rc=$?"; tail -4 $o/pytest_sas_solv.txt
for s in Euler RK4; do python3 bench.py --model sas --size 1000 100 --steps 6 --warmup 2 --sas-solver $s --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$s', d['ms_per_step'])"; done
