#!/bin/bash
# per-cell forcing: tests of the path, then ms per step / fused kernel per size with the default choice of front
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_hip_host_package.py tests/test_hip_loopback_ranks.py -x -q -m gpu -k "cell or station or weight or eberb or front or forcing" > gpurun_out/r4/cell_tests.log 2>&1; tail -5 gpurun_out/r4/cell_tests.log
one() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), round(d['roofline']['avg_kernel_ms'],4))" 2>&1; }
run() { python bench.py "$@" --no-cpu-baseline --no-extras 2>/dev/null; }
for i in 1 2; do
run --steps 200 --warmup 10 | one shared_1e6
run --station-weights --steps 200 --warmup 10 | one cell_1e6
run --size 80 53 --params hetero --steps 2000 --warmup 50 | one shared_80x53
run --size 80 53 --params hetero --station-weights --steps 2000 --warmup 50 | one cell_80x53
run --size 256 256 --station-weights --steps 1000 --warmup 50 | one cell_256
done
