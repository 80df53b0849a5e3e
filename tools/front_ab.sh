#!/bin/bash
# per-cell forcing in front of the fused step: the one-launch front against the predicate kernels, per grid size; optional variant library
#   gpurun -- tools/front_ab.sh [variant]
mkdir -p gpurun_out/r4
one() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), round(d['roofline']['avg_kernel_ms'],4))" 2>&1; }
run() { python bench.py "$@" --no-cpu-baseline --no-extras 2>/dev/null; }
for lib in product $1; do
  if [ $lib = product ]; then unset ROGER_HIP_LIB; else export ROGER_HIP_LIB=roger_amd/variants/libroger_hip_$lib.so; fi
  echo "== $lib"
  for i in 1 2; do
    run --steps 200 --warmup 10 | one shared_1e6
    run --station-weights --steps 200 --warmup 10 | one pred_1e6
    RH_CELL_FRONT_MAX=4000000 run --station-weights --steps 200 --warmup 10 | one front_1e6
    run --size 80 53 --params hetero --station-weights --steps 2000 --warmup 50 | one front_80x53
  done
done
