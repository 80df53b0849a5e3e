#!/bin/bash
# A/B of the parameter fast paths of the fused step inside ONE call (one box, alternating): the wave words switched off by environment
# (RH_NO_PARAM_UNIFORM / RH_NO_PARAM_DERIVE, read by rh_create) against the default, `rounds` times each.
#     gpurun -- tools/ab_params.sh "--steps 200 --warmup 10" 3 tag
args=${1:---steps 200 --warmup 10}
rounds=${2:-3}
tag=${3:-ab}
mkdir -p gpurun_out/ab_params
line() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; print('%-14s ms/step %.4f kernel %.4f ms  bytes/col %.0f  frac %.3f  value %.3e' % ('$2', d['ms_per_step'], r['avg_kernel_ms'], sum(r['algorithmic_bytes_per_cell'].values()), r['frac'], d['value']))"; }
for r in $(seq 1 $rounds); do
  python3 bench.py $args --no-cpu-baseline > gpurun_out/ab_params/${tag}_both_$r.json 2> gpurun_out/ab_params/${tag}_both_$r.err && line gpurun_out/ab_params/${tag}_both_$r.json "all three"
  RH_NO_PARAM_UNIFORM=1 RH_NO_PARAM_DERIVE=1 RH_NO_MASK_CONSTANT=1 python3 bench.py $args --no-cpu-baseline > gpurun_out/ab_params/${tag}_none_$r.json 2> gpurun_out/ab_params/${tag}_none_$r.err && line gpurun_out/ab_params/${tag}_none_$r.json "neither"
  RH_NO_PARAM_UNIFORM=1 python3 bench.py $args --no-cpu-baseline > gpurun_out/ab_params/${tag}_derive_$r.json 2> gpurun_out/ab_params/${tag}_derive_$r.err && line gpurun_out/ab_params/${tag}_derive_$r.json "no uniform"
  RH_NO_PARAM_DERIVE=1 python3 bench.py $args --no-cpu-baseline > gpurun_out/ab_params/${tag}_uniform_$r.json 2> gpurun_out/ab_params/${tag}_uniform_$r.err && line gpurun_out/ab_params/${tag}_uniform_$r.json "no derive"
  RH_NO_MASK_CONSTANT=1 python3 bench.py $args --no-cpu-baseline > gpurun_out/ab_params/${tag}_nomk_$r.json 2> gpurun_out/ab_params/${tag}_nomk_$r.err && line gpurun_out/ab_params/${tag}_nomk_$r.json "no mask const"
done
