#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py --station-weights at 10^6 and 80 x 53 columns: the kernels of a per-cell step
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out/r4
for cfg in "1e6|--steps 200 --warmup 10" "80x53|--size 80 53 --params hetero --steps 2000 --warmup 50"; do
  name=${cfg%%|*}; args=${cfg#*|}
  rm -rf gpurun_out/r4/prof_cell_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/prof_cell_$name -- python3 bench.py --station-weights $args --no-cpu-baseline --no-extras > gpurun_out/r4/prof_cell_$name.json 2>/dev/null
  echo "== $name: $(python3 -c "import json; d=json.loads(open('gpurun_out/r4/prof_cell_$name.json').read().strip().splitlines()[-1]); print('ms per step', round(d['ms_per_step'],4))")"
  f=$(find gpurun_out/r4/prof_cell_$name -name '*kernel_stats.csv' | head -1)
  python3 - $f <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:7]:
    print("  ", r['Name'][:72].ljust(72), r['Calls'].rjust(6), str(round(float(r['AverageNs'])/1e3,2)).rjust(8), "us")
PY
done
