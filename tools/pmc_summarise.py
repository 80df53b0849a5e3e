#!/usr/bin/env python3
"""Summarise the two PMC passes (FETCH_SIZE, WRITE_SIZE) of one kernel into a record of profiles/traffic.json.

    python tools/pmc_summarise.py FETCH_DIR WRITE_DIR N_CELLS KEY "KERNEL SUBSTRING" [CALIB_CELLS] [--out profiles/traffic.json] [--source TEXT]

Counter handling follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are collected in separate passes (TCC slots),
are reported in KiB, and on gfx950 must be calibrated for access shapes other than 16 B per lane -- here against k_calib_copy, which
moves a known byte count with the same 8-byte-per-lane accesses as k_step (FETCH_SIZE reads 1/2 there, WRITE_SIZE exactly).
profiles/traffic.json: {KEY: {"<n_cells>": record}} -- bench.py reports `roofline.traffic` only for a (kernel variant, column count)
that was profiled."""
import argparse
import csv
import glob
import json
import os

CALIB_PLANES = 96


def per_kernel(dirname, counter):
    rows = {}
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            rows.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("n_cells", type=int)
    ap.add_argument("key")
    ap.add_argument("kernel")
    ap.add_argument("calib_cells", type=int, nargs="?")
    ap.add_argument("--out", default="profiles/traffic.json")
    ap.add_argument("--source", default="")
    a = ap.parse_args()
    fetch, write = per_kernel(a.fetch_dir, "FETCH_SIZE"), per_kernel(a.write_dir, "WRITE_SIZE")
    key = lambda d, sub: next(k for k in d if sub in k)  # noqa: E731
    avg = lambda v: sum(v) / len(v)  # noqa: E731
    known = CALIB_PLANES * (a.calib_cells or a.n_cells) * 8
    f_cal, w_cal = avg(fetch[key(fetch, "k_calib_copy")]) * 1024, avg(write[key(write, "k_calib_copy")]) * 1024
    f_scale, w_scale = known / f_cal, known / w_cal
    ks_f, ks_w = fetch[key(fetch, a.kernel)], write[key(write, a.kernel)]
    fb, wb = avg(ks_f) * 1024 * f_scale, avg(ks_w) * 1024 * w_scale
    rec = {
        "n_cells": a.n_cells,
        "kernel": key(fetch, a.kernel),
        "hbm_bytes_per_launch": fb + wb,
        "fetch_bytes_per_launch": fb,
        "write_bytes_per_launch": wb,
        "bytes_per_cell": (fb + wb) / a.n_cells,
        "launches": {"fetch_pass": len(ks_f), "write_pass": len(ks_w)},
        "calibration": {"kernel": "k_calib_copy", "known_bytes_each_way": known, "FETCH_SIZE_raw_bytes": f_cal, "fetch_scale": f_scale,
                        "WRITE_SIZE_raw_bytes": w_cal, "write_scale": w_scale},
        "raw": {"FETCH_SIZE_KiB": avg(ks_f), "WRITE_SIZE_KiB": avg(ks_w)},
        "source": a.source or f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, {len(ks_f)} launches",
    }
    table = json.load(open(a.out)) if os.path.exists(a.out) else {}
    if "n_cells" in table:   # the round-1 layout (a single record)
        table = {}
    table.setdefault(a.key, {})[str(a.n_cells)] = rec
    json.dump(table, open(a.out, "w"), indent=1, sort_keys=True)
    print(a.key, a.n_cells, f"{rec['bytes_per_cell']:.1f} B per cell ({fb / a.n_cells:.1f} read + {wb / a.n_cells:.1f} written)")


if __name__ == "__main__":
    main()
