#!/bin/bash
o=gpurun_out/r3; mkdir -p $o
for cfg in "107 0" "259 0" "107 300" "259 300" "107 1000" "107 0"; do
  set -- $cfg
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --spinup $1 --prewarm-ms $2 --no-cpu-baseline > $o/warm_$1_$2.json 2> $o/warm_$1_$2.err
  python3 -c "
import json; d=json.loads(open('$o/warm_$1_$2.json').read().strip().splitlines()[-1])
print('spinup $1 prewarm $2: ms/step %.4f kernel %.4f' % (d['ms_per_step'], d['roofline']['avg_kernel_ms']), {k:(v['steps'], round(v['avg_kernel_ms'],4)) for k,v in d['config']['dt_classes'].items()}, d['clocks_mhz'])"
done
