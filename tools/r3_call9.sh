#!/bin/bash
# SAS: parity of the restructured eight-class kernel, then A/B old / new
o=gpurun_out/r3; mkdir -p $o
timeout -k 10 900 python3 -m pytest tests/test_hip_sas.py tests/test_hip_sas_bromide.py tests/test_hip_host_sas.py -x -q -m gpu > $o/pytest_sas_park.txt 2>&1; echo "pytest rc=$?"; tail -5 $o/pytest_sas_park.txt
rm -f gpurun_out/ab/*
tools/ab_variants.sh "--model sas --size 1000 100 --steps 20 --warmup 2" 3 sasold 2>&1 | tail -4
python3 tools/sas_tie_rate.py 2>&1 | grep "TIES sas\|TIES random"
