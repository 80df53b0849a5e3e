#!/bin/bash
o=gpurun_out/r3; mkdir -p $o
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $o/pytest_rhpow.txt 2>&1; echo "pytest rc=$?"; tail -8 $o/pytest_rhpow.txt
tools/ab_variants.sh "--steps 200 --warmup 20" 4 powlib 2>&1 | tail -3
tools/ab_variants.sh "--steps 100 --warmup 20 --size 3200 3125" 2 powlib 2>&1 | tail -3
