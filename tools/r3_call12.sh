#!/bin/bash
o=gpurun_out/r3; mkdir -p $o
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $o/pytest_gpu_tile256.txt 2>&1; echo "pytest rc=$?"; tail -n 4 $o/pytest_gpu_tile256.txt
rm -f gpurun_out/ab/*
tools/ab_variants.sh "--model oned --steps 200 --warmup 20" 2 tile64 2>&1 | tail -n 2
rm -f gpurun_out/ab/*
tools/ab_variants.sh "--model oned --routing --steps 200 --warmup 20" 2 tile64 2>&1 | tail -n 2
rm -f gpurun_out/ab/*
tools/ab_variants.sh "--size 80 53 --params hetero --steps 2000 --warmup 50" 2 tile64 2>&1 | tail -n 2
rm -f gpurun_out/ab/*
tools/ab_variants.sh "--steps 200 --warmup 20 --params hetero" 2 tile64 2>&1 | tail -n 2
