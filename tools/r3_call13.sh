#!/bin/bash
rm -f gpurun_out/ab/*
tools/ab_variants.sh "--steps 200 --warmup 20" 4 tile64 2>&1 | tail -n 2
rm -f gpurun_out/ab/*
tools/ab_variants.sh "--model oned --size 3200 3125 --steps 60 --warmup 10" 2 tile64 2>&1 | tail -n 2
rm -f gpurun_out/ab/*
tools/ab_variants.sh "--size 3200 3125 --steps 60 --warmup 10" 2 tile64 2>&1 | tail -n 2
rm -f gpurun_out/ab/*
tools/ab_variants.sh "--gpus 1 --steps 20 --warmup 5" 4 tile64 2>&1 | tail -n 2
