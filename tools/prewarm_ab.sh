#!/bin/bash
# A/B of the driver's bench command (20 steps, 5 warm-up steps) with and without the untimed pre-warm copy, alternating.
set -o pipefail
mkdir -p gpurun_out
line() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; print('$2', 'value %.3e ms_per_step %.4f kernel %.4f' % (d['value'], d['ms_per_step'], r['avg_kernel_ms']), d.get('clocks_mhz'))"; }
for i in 1 2 3 4; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/pw_a$i.out 2> gpurun_out/pw_a$i.err && line gpurun_out/pw_a$i.out "plain      " || exit 1
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --prewarm-ms 300 > gpurun_out/pw_b$i.out 2> gpurun_out/pw_b$i.err && line gpurun_out/pw_b$i.out "prewarm 300" || exit 1
done
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --prewarm-ms 1500 > gpurun_out/pw_c.out 2> gpurun_out/pw_c.err && line gpurun_out/pw_c.out "prewarm 1500"
python3 bench.py --gpus 1 --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/pw_d.out 2> gpurun_out/pw_d.err && line gpurun_out/pw_d.out "200 steps  "
