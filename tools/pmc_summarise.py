#!/usr/bin/env python3
"""Summarise the two PMC passes (FETCH_SIZE, WRITE_SIZE) into profiles/traffic.json.

Counter handling follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are collected
in separate passes (TCC slots), are reported in KiB, and on gfx950 must be calibrated for access
shapes other than 16 B per lane -- here against k_calib_copy, which moves a known byte count with
the same 8-byte-per-lane accesses as k_step.
"""
import csv
import glob
import json
import os
import sys

CALIB_PLANES = 96


def per_kernel(dirname, counter):
    rows = {}
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            rows.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return rows


def main(fetch_dir, write_dir, n_cells, out, kernel="k_step<2, false, true>", calib_cells=None):
    """kernel: substring of the kernel name to report; calib_cells: cells of the context the calibration copy ran
    on (default: n_cells)."""
    fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    key = lambda d, sub: next(k for k in d if sub in k)  # noqa: E731
    avg = lambda v: sum(v) / len(v)  # noqa: E731
    known = CALIB_PLANES * (calib_cells or n_cells) * 8
    f_cal, w_cal = avg(fetch[key(fetch, "k_calib_copy")]) * 1024, avg(write[key(write, "k_calib_copy")]) * 1024
    f_scale, w_scale = known / f_cal, known / w_cal
    ks_f = fetch[key(fetch, kernel)]
    ks_w = write[key(write, kernel)]
    fb, wb = avg(ks_f) * 1024 * f_scale, avg(ks_w) * 1024 * w_scale
    rec = {
        "n_cells": n_cells,
        "kernel": key(fetch, kernel),
        "hbm_bytes_per_launch": fb + wb,
        "fetch_bytes_per_launch": fb,
        "write_bytes_per_launch": wb,
        "bytes_per_cell": (fb + wb) / n_cells,
        "launches": {"fetch_pass": len(ks_f), "write_pass": len(ks_w)},
        "calibration": {
            "kernel": "k_calib_copy", "known_bytes_each_way": known,
            "FETCH_SIZE_raw_bytes": f_cal, "fetch_scale": f_scale,
            "WRITE_SIZE_raw_bytes": w_cal, "write_scale": w_scale,
        },
        "raw": {"k_step_FETCH_SIZE_KiB": avg(ks_f), "k_step_WRITE_SIZE_KiB": avg(ks_w)},
    }
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], *(sys.argv[5:6] or ["k_step<2, false, true>"]),
         *([int(sys.argv[6])] if len(sys.argv) > 6 else []))
