#!/bin/bash
# the tail of the fused kernel: tests of the summary path, its phases (measurement build), and ms per step of the product
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_hip_sparse.py tests/test_hip_host_package.py tests/test_hip_restart.py -x -q -m gpu > gpurun_out/r4/tail_tests.log 2>&1; tail -4 gpurun_out/r4/tail_tests.log
one() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), round(d['roofline']['avg_kernel_ms'],4))" 2>&1; }
run() { python bench.py "$@" --no-cpu-baseline --no-extras 2>/dev/null; }
for i in 1 2; do
run --steps 200 --warmup 10 | one shared_1e6
run --steps 20 --warmup 5 | one shared_1e6_20
run --size 80 53 --params hetero --steps 2000 --warmup 50 | one shared_80x53
run --model oned --size 3200 3125 --steps 20 --warmup 5 | one oned_1e7
done
export ROGER_HIP_LIB=roger_amd/variants/libroger_hip_phases.so
python bench.py --size 80 53 --params hetero --steps 2000 --warmup 50 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep "tail phases"
python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep "tail phases"
