#!/bin/bash
# Rebuilds every number under profiles/ from ONE call on the GPU box, so that they cannot drift apart:
#     gpurun --timeout 1200 -- tools/make_profiles.sh r02      (then, in the build container:  cp gpurun_out/profiles/* profiles/)
# Bench lines (default, dt classes, 10^7, oneD, 80 x 53, SAS), rocprofv3 --kernel-trace --stats of the default bench command and of the
# SAS bench, PMC traffic (FETCH_SIZE / WRITE_SIZE in separate passes, calibrated) of every benched kernel variant and size, and the VALU
# instruction counters of the SAS kernel.  rocprofv3 is always given the program itself (python3 ...), never a wrapper.
set -o pipefail
tag=${1:-r04}
export RH_TAG=$tag
what=${2:-all}
out=gpurun_out/profiles
scratch=gpurun_out/prof_scratch
mkdir -p $out $scratch
# records of kernels / sizes that this call does not measure again stay as they are
for f in traffic.json sas_valu.json; do [ -f $out/$f ] || cp profiles/$f $out/$f 2> /dev/null; done
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export PYTHONUNBUFFERED=1
last() { tail -n 1 "$1"; }

bench() {   # name, args...
  local name=$1; shift
  python3 bench.py "$@" > $scratch/$name.out 2> $scratch/$name.err && last $scratch/$name.out > $out/${tag}_bench_$name.json && echo "bench $name: $(python3 -c "import json,sys; d=json.load(open('$out/${tag}_bench_$name.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['frac'])")"
}
pmc_pair() {   # key, kernel substring, n_cells, calib_cells, program args... (environment already exported)
  local key=$1 kern=$2 n=$3 calib=$4; shift 4
  rm -rf $scratch/pmc_f $scratch/pmc_w
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $scratch/pmc_f -- "$@" > $scratch/pmc_f.log 2>&1 || { echo "pmc fetch pass failed: $key"; tail -3 $scratch/pmc_f.log; return 1; }
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $scratch/pmc_w -- "$@" > $scratch/pmc_w.log 2>&1 || { echo "pmc write pass failed: $key"; tail -3 $scratch/pmc_w.log; return 1; }
  python3 tools/pmc_summarise.py $scratch/pmc_f $scratch/pmc_w $n "$key" "$kern" $calib --out $out/traffic.json --source "tools/make_profiles.sh $tag: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), calibrated on k_calib_copy"
}

if [ "$what" = all ] || [ "$what" = svat ]; then
  # the sparse variant (every step of an rh_run_steps call but the last) and, with RH_NO_SPARSE_STORES=1, the full-store lazy variant
  svat_pairs() {   # model, size, n
    export RH_PMC_MODEL=$1 RH_PMC_SIZE=$2
    local lat=false; [ $1 = oned ] && lat=true
    pmc_pair k_step_$1_sparse "k_step<2, $lat, true, true, false>" $3 $3 python3 tools/pmc_workload.py
    RH_NO_SPARSE_STORES=1 pmc_pair k_step_$1_lazy "k_step<2, $lat, true, false, false>" $3 $3 python3 tools/pmc_workload.py
  }
  svat_pairs svat 1000x1000 1000000
  # heterogeneous parameters (drawn per column: no wave reads a parameter plane as one element; the derived parameters are still not loaded)
  RH_PMC_PARAMS=hetero RH_PMC_MODEL=svat RH_PMC_SIZE=1000x1000 pmc_pair k_step_svat_sparse_hetero "k_step<2, false, true, true, false>" 1000000 1000000 python3 tools/pmc_workload.py
  svat_pairs oned 1000x1000 1000000
  svat_pairs svat 3200x3125 10000000
  svat_pairs oned 3200x3125 10000000
  svat_pairs svat 80x53 4240
  unset RH_PMC_MODEL RH_PMC_SIZE
  cp $out/traffic.json profiles/   # the bench lines take roofline.traffic from the records just written
  bench svat_1e6 --steps 200 --warmup 10
  bench svat_1e6_default
  bench svat_1e6_dt_classes --steps 600 --warmup 10 --no-cpu-baseline
  bench svat_1e6_hetero --params hetero --steps 200 --warmup 10 --no-cpu-baseline
  bench svat_1e7 --size 3200 3125 --steps 60 --warmup 5 --no-cpu-baseline
  bench oned_1e6 --model oned --steps 200 --warmup 10 --no-cpu-baseline
  bench oned_1e7 --model oned --size 3200 3125 --steps 60 --warmup 5 --no-cpu-baseline
  bench svat_80x53 --size 80 53 --params hetero --steps 2000 --warmup 50 --no-cpu-baseline
  bench svat_80x53_station_weights --size 80 53 --params hetero --station-weights --steps 2000 --warmup 50 --no-cpu-baseline
  bench svat_1e6_station_weights --station-weights --steps 200 --warmup 10 --no-cpu-baseline
  RH_BENCH_FORCE_DIST=1 bench svat_1e6_rccl_one_rank --steps 200 --warmup 10 --no-cpu-baseline
  bench svat_1e6_driver_command --gpus 1 --steps 20 --warmup 5
  RH_NO_SPARSE_STORES=1 bench svat_1e6_full_stores --steps 200 --warmup 10 --no-cpu-baseline
  # the benchmark as a RogerSetup script calling plain run(): stock hooks (on the device), a hook of the script's own in front of the physics (one native call per step) and the three-call step
  for st in setup script hooks routines; do
    python3 bench.py --stepping $st --days 20 --warmup-days 2 > $scratch/svat_1e6_$st.out 2> $scratch/svat_1e6_$st.err && last $scratch/svat_1e6_$st.out > $out/${tag}_bench_svat_1e6_run_$st.json && echo "run() $st ok"
    python3 bench.py --stepping $st --size 80 53 --days 60 --warmup-days 2 > $scratch/svat_80x53_$st.out 2> $scratch/svat_80x53_$st.err && last $scratch/svat_80x53_$st.out > $out/${tag}_bench_svat_80x53_run_$st.json
  done
  # the parameter fast paths of the fused step, alternating in this very call
  {
    echo "# The parameter fast paths of the fused step switched off one at a time (tools/ab_params.sh: RH_NO_PARAM_UNIFORM / RH_NO_PARAM_DERIVE /"
    echo "# RH_NO_MASK_CONSTANT), alternating inside this one call; bytes/col = what the variant must move (census x rh_param_stats)."
    echo "# Blocks: 10^6 columns uniform parameters (two rounds), 10^6 columns parameters drawn per column, 10^7 columns uniform."
    echo "# Earlier in round 4 (removed from the source): the loads of stage k + 2 requested before stage k -- everywhere: + 5 % and spills;"
    echo "# through the three short early stages only: +- 1 %."
  } > $out/${tag}_ab_param_paths.txt
  tools/ab_params.sh "--steps 200 --warmup 10" 2 ${tag}_u1e6 >> $out/${tag}_ab_param_paths.txt 2>&1
  tools/ab_params.sh "--steps 200 --warmup 10 --params hetero" 1 ${tag}_h1e6 >> $out/${tag}_ab_param_paths.txt 2>&1
  tools/ab_params.sh "--steps 60 --warmup 5 --size 3200 3125" 1 ${tag}_u1e7 >> $out/${tag}_ab_param_paths.txt 2>&1
  # BASELINE configs[4] as written: the Eberbaechle shape over the station's full 2019-2022 series
  python3 tools/eberbaechle_full.py > $out/${tag}_eberbaechle_full_80x53.json 2> $scratch/eberbaechle_full.err && echo "eberbaechle full ok"
  # what bounds the fused step: SQ counters per launch of every k_step variant
  tools/svat_pmc.sh svat 1000x1000 > $out/${tag}_svat_sq_counters_1e6.txt 2>&1 && echo "sq counters ok"
  # the default bench command under rocprofv3 --kernel-trace --stats: the kernel's average duration there must agree with the HIP events
  rm -rf $scratch/stats_svat
  rocprofv3 --kernel-trace --stats --output-format csv -d $scratch/stats_svat -- python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline > $scratch/stats_svat.out 2> $scratch/stats_svat.err \
    && last $scratch/stats_svat.out > $out/${tag}_bench_svat_1e6_under_rocprof.json \
    && cp "$(find $scratch/stats_svat -name '*kernel_stats.csv' | head -1)" $out/${tag}_kernel_stats_svat_1e6.csv && echo "kernel stats svat ok"
fi
if [ "$what" = all ] || [ "$what" = routing ]; then
  # the routed step (settings.enable_routing_1D): traffic of its longest pass, the bench line, the per-kernel times of one step
  export RH_PMC_MODEL=routed RH_PMC_SIZE=1000x1000
  pmc_pair k_routed_a2 "k_routed_a2" 1000000 1000000 python3 tools/pmc_workload.py
  unset RH_PMC_MODEL RH_PMC_SIZE
  cp $out/traffic.json profiles/
  bench oned_1e6_routing --model oned --routing --steps 200 --warmup 10 --no-cpu-baseline
  rm -rf $scratch/stats_routed
  rocprofv3 --kernel-trace --stats --output-format csv -d $scratch/stats_routed -- python3 bench.py --model oned --routing --steps 200 --warmup 10 --no-cpu-baseline > $scratch/stats_routed.out 2> $scratch/stats_routed.err \
    && last $scratch/stats_routed.out > $out/${tag}_bench_oned_1e6_routing_under_rocprof.json \
    && cp "$(find $scratch/stats_routed -name '*kernel_stats.csv' | head -1)" $out/${tag}_kernel_stats_oned_1e6_routing.csv && echo "kernel stats routed ok"
fi
if [ "$what" = all ] || [ "$what" = sas ]; then
  export RH_PMC_CALIB=1
  pmc_pair k_sas_ages1000_sub6 "k_sas8<" 100000 1000000 python3 tools/sas_workload.py 100000 4
  pmc_pair k_sas_euler_ages1000_sub6 "k_sas_euler<" 100000 1000000 python3 tools/sas_workload.py 100000 3 6 1 Euler
  unset RH_PMC_CALIB
  # VALU issue: wave-instructions of the SAS kernels per launch (SQ block, one pass), and the busy cycles beside them
  valu() {   # key, workload args...
    local key=$1; shift
    rm -rf $scratch/pmc_valu
    rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $scratch/pmc_valu -- python3 tools/sas_workload.py "$@" > $scratch/pmc_valu.log 2>&1 \
      && python3 tools/sas_valu_summarise.py $scratch/pmc_valu 100000 $out/sas_valu.json "tools/make_profiles.sh $tag: rocprofv3 --pmc SQ_INSTS_VALU ... on tools/sas_workload.py $*" $key \
      && cp "$(find $scratch/pmc_valu -name '*counter_collection.csv' | head -1)" $scratch/sas_counters_$key.csv
  }
  valu ages1000_sub6 100000 4
  valu euler_ages1000_sub6 100000 3 6 1 Euler
  valu rk4_ages1000_sub6 100000 2 6 1 RK4
  # the bench lines take roofline.traffic / roofline.compute from the records just written
  cp $out/traffic.json $out/sas_valu.json profiles/
  bench sas_1e6 --model sas --steps 8 --warmup 2
  # 10^5 columns: the size the counters below are collected at (bench.py fills roofline.traffic / roofline.compute from them on a later run)
  bench sas_1e5 --model sas --size 1000 100 --steps 20 --warmup 2 --no-cpu-baseline
  bench sas_1e5_euler --model sas --size 1000 100 --steps 10 --warmup 2 --sas-solver Euler --no-cpu-baseline
  bench sas_1e5_rk4 --model sas --size 1000 100 --steps 6 --warmup 2 --sas-solver RK4 --no-cpu-baseline
  rm -rf $scratch/stats_sas
  rocprofv3 --kernel-trace --stats --output-format csv -d $scratch/stats_sas -- python3 bench.py --model sas --size 1000 100 --steps 8 --warmup 2 --no-cpu-baseline > $scratch/stats_sas.out 2> $scratch/stats_sas.err \
    && last $scratch/stats_sas.out > $out/${tag}_bench_sas_1e5_under_rocprof.json \
    && cp "$(find $scratch/stats_sas -name '*kernel_stats.csv' | head -1)" $out/${tag}_kernel_stats_sas_1e5.csv && echo "kernel stats sas ok"
  python3 - <<'PY'
import csv, collections, glob, os
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob("gpurun_out/prof_scratch/sas_counters_*.csv")):
    for r in csv.DictReader(open(path)):
        rows[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/profiles/TAG_pmc_sas_valu_by_kernel.csv".replace("TAG", os.environ.get("RH_TAG", "r02")), "w") as f:
    f.write("kernel,counter,launches,average\n")
    for k, c in rows.items():
        for name, v in sorted(c.items()):
            f.write(f"\"{k}\",{name},{len(v)},{sum(v) / len(v):.1f}\n")
PY
fi
ls -la $out
