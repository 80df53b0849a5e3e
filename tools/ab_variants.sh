#!/bin/bash
# A/B of kernel variants (tools/build_variant.py) on the GPU box: alternates the product library and the variants, prints ms per step
# and the fused kernel's average duration.   usage: tools/ab_variants.sh "<bench args>" rounds name1 name2 ...
args="$1"; rounds="$2"; shift 2
mkdir -p gpurun_out/ab
for r in $(seq 1 "$rounds"); do
  for v in base "$@"; do
    if [ "$v" = base ]; then lib=roger_amd/libroger_hip.so; else lib=roger_amd/variants/libroger_hip_$v.so; fi
    RH_OLD_VARIANT=1 ROGER_HIP_LIB=$lib python bench.py $args --no-cpu-baseline > gpurun_out/ab/${v}_$r.json 2> gpurun_out/ab/${v}_$r.err || echo "$v failed"
  done
done
python - "$@" <<'PY'
import glob, json, statistics, sys
for v in ["base"] + sys.argv[1:]:
    rows = []
    for f in sorted(glob.glob(f"gpurun_out/ab/{v}_*.json")):
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1])
            rows.append((d["ms_per_step"], d["roofline"]["avg_kernel_ms"]))
        except Exception as e:
            print(f, "unreadable", e)
    if rows:
        print(f"{v:10s} ms/step " + " ".join(f"{a:.4f}" for a, _ in rows) + f" | kernel " + " ".join(f"{b:.4f}" for _, b in rows) +
              f" | median step {statistics.median(a for a, _ in rows):.4f} kernel {statistics.median(b for _, b in rows):.4f}")
PY
