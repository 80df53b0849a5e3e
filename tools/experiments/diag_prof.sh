python3 tools/diag_time.py 300 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out/r4
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/prof_diag -- python3 tools/diag_time.py 300 > /dev/null 2>&1
f=$(find gpurun_out/r4/prof_diag -name '*kernel_stats.csv' | head -1); python3 - $f <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    print(r['Name'][:70].ljust(70), r['Calls'], round(float(r['AverageNs'])/1e3,2))
PY
