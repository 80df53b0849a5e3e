#!/bin/bash
o=gpurun_out/r3; mkdir -p $o
ROGER_HIP_LIB=roger_amd/variants/libroger_hip_pf2.so timeout -k 10 600 python3 -m pytest tests/test_hip_parity.py tests/test_hip_sparse.py -x -q -m gpu -k "trajectory_golden or pieces or device_hooks_golden or lazy or weighted_station_forcing_golden or time_limit or eberbaechle" > $o/pytest_pf2.txt 2>&1; echo "pytest pf2 rc=$?"; tail -2 $o/pytest_pf2.txt
tools/ab_variants.sh "--steps 200 --warmup 20" 5 pf2 2>&1 | tail -3
tools/ab_variants.sh "--steps 100 --warmup 20 --size 3200 3125" 2 pf2 2>&1 | tail -3
tools/ab_variants.sh "--steps 200 --warmup 20 --model oned" 2 pf2 2>&1 | tail -3
