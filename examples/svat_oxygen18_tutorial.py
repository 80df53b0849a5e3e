#!/usr/bin/env python3
"""SVAT -> oxygen-18 transport, the reference's two-step workflow (examples/plot_scale/svat_tutorial followed by
svat_oxygen18_tutorial) on the hip backend:

    python examples/svat_oxygen18_tutorial.py /path/to/svat_tutorial/input [--days 365] [--ages 400] [--out output]

1. the SVAT tutorial (examples/svat_tutorial.py) writes the daily flux sums and storages (SVAT.rate.nc, SVAT.collect.nc);
2. `read_svat_output` turns them into the transport model's input; the isotope signal of the precipitation is synthetic
   here (a seasonal cycle around -8 permil: the tutorial ships no measured d18O);
3. the transport model (roger_amd/models/svat_transport.py; power-law StorAge selection with the benchmark's exponents,
   deterministic solver) warms up over the whole period once, rescales, and runs; it writes the daily signal of
   transpiration and percolation and their median travel times as SVATOXYGEN18.average.nc.
"""
import argparse
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from roger_amd import roger_routine  # noqa: E402
from roger_amd.models.svat_transport import make_transport_model, read_svat_output  # noqa: E402
from roger_amd.sas import benchmark_sas_params  # noqa: E402


def _svat_example():
    spec = importlib.util.spec_from_file_location("svat_tutorial_example", os.path.join(HERE, "svat_tutorial.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("input_dir")
    ap.add_argument("--days", type=int, default=365)
    ap.add_argument("--ages", type=int, default=400)
    ap.add_argument("--substeps", type=int, default=6)
    ap.add_argument("--out", default="output")
    args = ap.parse_args(argv)
    _svat_example().main([args.input_dir, "--days", str(args.days), "--out", args.out])
    svat = read_svat_output(os.path.join(args.out, "SVAT.rate.nc"), os.path.join(args.out, "SVAT.collect.nc"))
    doy = np.arange(args.days + 1)
    svat["d18O_prec"] = np.concatenate([[np.nan], (-8.0 + 3.0 * np.sin(2 * np.pi * (doy[1:] - 30) / 365.25))])
    nx, ny = svat["S_rz"].shape[:2]
    sas = {f: p.reshape(nx, ny, 8) for f, p in benchmark_sas_params(nx * ny).items()}
    base = make_transport_model("roger_amd", svat, sas, args.ages, args.substeps, args.days, age_statistics=True,
                                warmup_days=args.days)

    class WithOutput(type(base)):
        @roger_routine
        def set_settings(self, state):
            super().set_settings(state)
            state.settings.identifier = "SVATOXYGEN18"

        @roger_routine
        def set_diagnostics(self, state):
            d = state.diagnostics["average"]
            d.output_variables = ["C_iso_transp", "C_iso_s", "C_iso_q_ss", "tt50_transp", "rt50_s", "tt50_q_ss"]   # the tutorial's OUTPUT_AVERAGE
            d.output_frequency = 24 * 60 * 60
            d.sampling_frequency = 1
            d.base_output_path = args.out

    model = WithOutput()
    model.setup()
    model.warmup()
    model.run()
    vs = model.state.variables
    print(f"transport: {vs.itt} days; d18O of the percolation on the last day {np.asarray(vs.C_iso_q_ss)[2:-2, 2:-2].ravel()} permil; "
          f"output in {args.out}/SVATOXYGEN18.average.nc")
    return model


if __name__ == "__main__":
    main()
