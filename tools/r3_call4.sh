#!/bin/bash
o=gpurun_out/r3; mkdir -p $o
for v in tile256 tile128; do
  ROGER_HIP_LIB=roger_amd/variants/libroger_hip_$v.so timeout -k 10 600 python3 -m pytest tests/test_hip_parity.py tests/test_hip_sparse.py tests/test_hip_routing.py -x -q -m gpu -k "trajectory_golden or pieces or device_hooks_golden or lazy or cut or weighted_station_forcing_golden or full_size" > $o/pytest_$v.txt 2>&1; echo "pytest $v rc=$?"; tail -2 $o/pytest_$v.txt
done
tools/ab_variants.sh "--steps 200 --warmup 20" 3 tile256 tile128 2>&1 | tail -5
tools/ab_variants.sh "--steps 100 --warmup 20 --size 3200 3125" 2 tile256 2>&1 | tail -4
