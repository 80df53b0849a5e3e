#!/bin/bash
# Where a wavefront of the fused step spends its cycles, stage by stage (a measurement build: -DRH_STEP_PHASES adds a clock read and an
# atomic per stage and wave, + 10 % on the step).   gpurun -- tools/step_phases.sh ["<bench args>"]
#   python tools/build_variant.py phases --unit roger_hip -DRH_STEP_PHASES      (first, in the build container)
# stages: 1 prec/ta + pet selection (and the first loads), 2 interception, 3 evapotranspiration, 4 snow, 5 infiltration events, 6 matrix
# infiltration, 7 macropores, 8 cracks, 9 infiltration finish, 10 subsurface runoff (with k / h of both layers from the water contents),
# 11 capillary rise, 12 storages, 13 numerical error, 14 after_timestep
args="${1:---steps 200 --warmup 10}"
ROGER_HIP_LIB=roger_amd/variants/libroger_hip_phases.so python bench.py $args --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep "step phases"
