for r in 1 2; do for v in "" notail nocompl; do if [ -z "$v" ]; then unset ROGER_HIP_LIB; else export ROGER_HIP_LIB=roger_amd/variants/libroger_hip_$v.so; fi; python tools/experiments/time_steps.py 1000 1000 200 2>&1 | tail -1; python tools/experiments/time_steps.py 80 53 2000 hetero 2>&1 | tail -1; done; done
unset ROGER_HIP_LIB
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out/r4
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/prof_80w -- python3 bench.py --size 80 53 --params hetero --station-weights --steps 2000 --warmup 50 --no-cpu-baseline --no-extras > /dev/null 2>&1
f=$(find gpurun_out/r4/prof_80w -name '*kernel_stats.csv' | head -1); python3 - $f <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    print(r['Name'][:60].ljust(60), r['Calls'], round(float(r['AverageNs'])/1e3,2))
PY
