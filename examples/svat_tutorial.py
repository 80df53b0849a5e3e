#!/usr/bin/env python3
"""The reference's plot-scale tutorial (examples/plot_scale/svat_tutorial: one grassland cell, a year of measured
10-minute precipitation and daily temperature / potential evapotranspiration) on the hip backend.

    python examples/svat_tutorial.py /path/to/svat_tutorial/input [--days 365] [--out OUTPUT_DIR]

reads PREC.txt / TA.txt / PET.txt (roger_amd.forcing.forcing_from_txt: the reference's read_meteo + write_forcing recipe),
steps the year on the GPU with the forcing hooks on the device, and writes the daily sums of the fluxes and the
end-of-day storages as <identifier>.rate.nc / <identifier>.collect.nc in the reference's layout (roger_amd.diagnostics).
Parameters: the tutorial's config.yml.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from roger_amd import roger_routine  # noqa: E402
from roger_amd.core.operators import at, update  # noqa: E402
from roger_amd.forcing import forcing_from_txt  # noqa: E402
from roger_amd.models.svat import SVATSetup  # noqa: E402

CONFIG = dict(LU_ID=8, SEALING=0, Z_SOIL=900, DMPV=0, LMPV=500, THETA_AC=0.1, THETA_UFC=0.1, THETA_PWP=0.2, KS=10, KF=2500,
              THETA_RZ=0.3, THETA_SS=0.3, OUTPUT_FREQUENCY=86400,
              OUTPUT_RATE=["prec", "aet", "transp", "evap_soil", "inf_mat_rz", "inf_mp_rz", "inf_sc_rz", "inf_ss", "q_rz",
                           "q_ss", "cpr_rz", "q_hof", "q_sof"],
              OUTPUT_COLLECT=["S_rz", "S_ss", "S_pwp_rz", "S_pwp_ss", "S_sat_rz", "S_sat_ss", "theta", "S_snow", "S"])


def make_model(input_dir, ndays, out_dir):
    forcing = forcing_from_txt(input_dir, ndays=ndays)

    class Tutorial(SVATSetup):
        @roger_routine
        def set_settings(self, state):
            s = state.settings
            s.identifier = "SVAT"
            s.output_frequency = CONFIG["OUTPUT_FREQUENCY"]
            s.nx, s.ny = 1, 1
            s.runlen = 86400 * ndays
            s.nitt_forc = len(forcing["PREC"])
            s.dx = s.dy = 1
            s.x_origin = s.y_origin = 0.0
            s.time_origin = "2010-09-30 23:00:00"     # the tutorial's forcing starts on 1 October 2010
            s.enable_groundwater_boundary = False
            s.enable_macropore_lower_boundary_condition = False
            s.enable_adaptive_time_stepping = True

        @roger_routine
        def set_parameters_setup(self, state):
            vs = state.variables
            for name, key in (("lu_id", "LU_ID"), ("sealing", "SEALING"), ("z_soil", "Z_SOIL"), ("dmpv", "DMPV"),
                              ("lmpv", "LMPV"), ("theta_ac", "THETA_AC"), ("theta_ufc", "THETA_UFC"),
                              ("theta_pwp", "THETA_PWP"), ("ks", "KS"), ("kf", "KF")):
                setattr(vs, name, update(getattr(vs, name), at[2:-2, 2:-2], CONFIG[key]))

        @roger_routine
        def set_initial_conditions(self, state):
            vs = state.variables
            vs.theta_rz = update(vs.theta_rz, at[2:-2, 2:-2, : vs.taup1], CONFIG["THETA_RZ"])
            vs.theta_ss = update(vs.theta_ss, at[2:-2, 2:-2, : vs.taup1], CONFIG["THETA_SS"])

        @roger_routine
        def set_diagnostics(self, state):
            d = state.diagnostics
            for kind, key in (("rate", "OUTPUT_RATE"), ("collect", "OUTPUT_COLLECT")):
                d[kind].output_variables = CONFIG[key]
                d[kind].output_frequency = CONFIG["OUTPUT_FREQUENCY"]
                d[kind].sampling_frequency = 1
                d[kind].base_output_path = out_dir

    return Tutorial(forcing=forcing, nx=1, ny=1, ndays=ndays)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("input_dir", help="directory with PREC.txt, TA.txt, PET.txt")
    ap.add_argument("--days", type=int, default=365)
    ap.add_argument("--out", default="output")
    args = ap.parse_args(argv)
    os.makedirs(args.out, exist_ok=True)
    model = make_model(args.input_dir, args.days, args.out)
    model.setup()
    vs = model.state.variables
    end = args.days * 86400
    while vs.time < end:       # hooks on the device; a step covers at most a day, so 64 steps cannot pass the end from 65 days out
        model.run_device(64 if end - vs.time > 65 * 86400 else 1)
    print(f"{vs.itt} steps, {vs.time // 86400} days; output in {args.out}/SVAT.rate.nc, SVAT.collect.nc")
    return model


if __name__ == "__main__":
    main()
