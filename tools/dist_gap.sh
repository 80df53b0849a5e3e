#!/bin/bash
# Where the multi-GPU step's time goes on one GPU: the fused step with its control part in the kernel's tail, with k_ctrl as a kernel of
# its own, and rh_run_steps_dist with a one-rank communicator (RCCL all-reduce / a plain copy in its place).
#     gpurun -- tools/dist_gap.sh
set -o pipefail
mkdir -p gpurun_out
line() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; print('$2', 'ms_per_step %.4f kernel %.4f outside_us %.1f' % (d['ms_per_step'], r['avg_kernel_ms'], r.get('outside_kernel_us_per_step', 0)))"; }
python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/gap_a.out 2> gpurun_out/gap_a.err && line gpurun_out/gap_a.out "tail control        " \
&& RH_NO_TAIL_CTRL=1 python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/gap_d.out 2> gpurun_out/gap_d.err && line gpurun_out/gap_d.out "k_ctrl + k_step     " \
&& RH_BENCH_FORCE_DIST=1 python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/gap_b.out 2> gpurun_out/gap_b.err && line gpurun_out/gap_b.out "dist, RCCL 1 rank   " \
&& RH_BENCH_FORCE_DIST=1 RH_DIST_COPY_ONLY=1 python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/gap_c.out 2> gpurun_out/gap_c.err && line gpurun_out/gap_c.out "dist, copy instead  " \
&& rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gap_stats -- python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/gap_s.out 2> gpurun_out/gap_s.err \
&& RH_BENCH_FORCE_DIST=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gap_stats_dist -- python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/gap_sd.out 2> gpurun_out/gap_sd.err \
&& head -8 $(find gpurun_out/gap_stats_dist -name '*kernel_stats.csv' | head -1) | cut -c1-140
