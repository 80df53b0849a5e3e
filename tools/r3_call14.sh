#!/bin/bash
o=gpurun_out/r3; mkdir -p $o
timeout -k 10 900 python3 -m pytest tests/test_hip_sas.py tests/test_hip_sas_bromide.py tests/test_hip_host_sas.py -x -q -m gpu > $o/pytest_sas_fin.txt 2>&1; echo "pytest rc=$?"; tail -n 3 $o/pytest_sas_fin.txt
rm -f gpurun_out/ab/*
tools/ab_variants.sh "--model sas --size 1000 100 --steps 20 --warmup 2" 3 spill 2>&1 | tail -n 2
rm -f gpurun_out/ab/*
tools/ab_variants.sh "--model sas --size 1000 100 --steps 6 --warmup 2 --sas-solver RK4" 2 spill 2>&1 | tail -n 2
