#!/bin/bash
# Round 3: which arena layout is insensitive to the allocation's placement?   gpurun -- tools/placement_experiment.sh
# (build first: hipcc --offload-arch=gfx950 -O3 tools/experiments/placement_probe.hip -o roger_amd/variants/placement_probe;
#  python tools/build_variant.py planemajor --unit roger_hip -DRH_TILED=0)
o=gpurun_out/placement
mkdir -p $o
for a in malloc contig carve; do
  timeout -k 10 120 roger_amd/variants/placement_probe 8 $a 267 > $o/probe_$a.txt 2>&1 || echo "probe $a failed"
done
timeout -k 10 120 roger_amd/variants/placement_probe 8 malloc 276 > $o/probe_malloc_276.txt 2>&1 || echo "probe 276 failed"
timeout -k 10 300 python3 tools/arena_levels.py 8 > $o/levels_tiled.txt 2>&1 || echo "levels tiled failed"
ROGER_HIP_LIB=roger_amd/variants/libroger_hip_planemajor.so timeout -k 10 300 python3 tools/arena_levels.py 8 > $o/levels_planemajor.txt 2>&1 || echo "levels planemajor failed"
tail -n 30 $o/probe_malloc.txt $o/levels_tiled.txt $o/levels_planemajor.txt
