#!/usr/bin/env python3
"""VALU issue census of the SAS kernel from one rocprofv3 --pmc pass (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, SQ_WAVE_CYCLES, SQ_BUSY_CYCLES,
GRBM_GUI_ACTIVE) -> profiles/sas_valu.json, the input of bench.py's `roofline.compute` for --model sas.

    python tools/sas_valu_summarise.py PMC_DIR N_CELLS OUT SOURCE_TEXT [KEY]     (KEY: ages1000_sub6, euler_ages1000_sub6, ...)

SQ_INSTS_VALU counts wave-instructions summed over the chip; an fp64 wave64 VALU instruction occupies its SIMD's 16-lane fp64 pipe for
4 cycles, the chip has 256 CUs x 4 SIMDs: the kernel cannot take less than INSTS x 4 / 1024 cycles (MI355X_MICROARCH.md)."""
import csv
import glob
import json
import os
import sys


def main(pmc_dir, n_cells, out, source, key="ages1000_sub6"):
    vals, dur = {}, []
    for f in glob.glob(os.path.join(pmc_dir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_sas" in r["Kernel_Name"]:
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                kern = r["Kernel_Name"]
                if r["Counter_Name"] == "SQ_INSTS_VALU":
                    dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    avg = {k: sum(v) / len(v) for k, v in vals.items()}
    insts = avg["SQ_INSTS_VALU"]
    rec = {"n_cells": n_cells, "kernel": kern, "launches": len(vals["SQ_INSTS_VALU"]), "counters_per_launch": avg,
           "valu_wave_insts_per_column": insts / n_cells, "clock_mhz": 2400, "source": source}
    rec["kernel_ms_under_pmc"] = sum(dur) / len(dur) / 1e6
    rec["min_cycles_at_full_issue"] = insts * 4 / 1024
    if "GRBM_GUI_ACTIVE" in avg:
        rec["gui_active_cycles"] = avg["GRBM_GUI_ACTIVE"]
        rec["issue_fraction_by_counters"] = insts * 4 / 1024 / (avg["GRBM_GUI_ACTIVE"] / 8)   # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        rec["clock_mhz"] = avg["GRBM_GUI_ACTIVE"] / 8 / (sum(dur) / len(dur)) * 1e3   # cycles per ns -> MHz
    table = json.load(open(out)) if os.path.exists(out) else {}
    table[key] = rec
    json.dump(table, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(rec))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4], *sys.argv[5:6])
