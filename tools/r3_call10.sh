#!/bin/bash
o=gpurun_out/r3; mkdir -p $o
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $o/pytest_gpu_r5.txt 2>&1; echo "pytest rc=$?"; tail -6 $o/pytest_gpu_r5.txt
