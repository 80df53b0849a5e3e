#!/usr/bin/env python3
"""The reference's hillslope routing example (examples/hillslope_scale/oneD_distributed_routing_tutorial: a 1 x 20 hillslope whose cells
drain into the next one downslope, `settings.enable_routing_1D`) on the hip backend.

    python examples/oned_routing_tutorial.py /path/to/oneD_distributed_routing_tutorial/input [--days 30] [--out OUTPUT_DIR]

reads PREC.txt / TA.txt / PET.txt (roger_amd.forcing.forcing_from_txt), steps on the GPU with the forcing hooks on the device
(`run_device`: the routed step, three per-column passes around the two D8 gathers) and writes the daily sums of the fluxes and the
end-of-day storages as ONED.rate.nc / ONED.collect.nc.  Grid, parameters and initial state: the example's config.yml and oneD.py:127-335,
455-483; the station weights of its set_forcing (precipitation x 1, air temperature + 1, potential evapotranspiration x 1) go to the
device with the series (rh_set_forcing_weights)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from roger_amd import roger_routine  # noqa: E402
from roger_amd.core.operators import at, update  # noqa: E402
from roger_amd.forcing import forcing_from_txt  # noqa: E402
from roger_amd.models.oned import ONEDSetup  # noqa: E402

NX, NY = 1, 20
PARAMETERS = dict(lu_id=8, sealing=0, S_dep_tot=0, z_soil=900, slope=0.05, dmpv=50, dmph=50, lmpv=600, theta_ac=0.1, theta_ufc=0.1,
                  theta_pwp=0.2, ks=20, kf=1, k_st=50, prec_weight=1, ta_offset=1, pet_weight=1)
OUTPUT_RATE = ["prec", "aet", "transp", "evap_soil", "inf_mat_rz", "inf_mp_rz", "inf_sc_rz", "inf_ss", "q_rz", "q_ss", "cpr_rz", "q_sub",
               "q_sur_out", "q_sur_in", "q_sub_out", "q_sub_in"]
OUTPUT_COLLECT = ["S_rz", "S_ss", "S", "z0", "z_sat", "theta"]


def make_model(input_dir, ndays, out_dir):
    forcing = forcing_from_txt(input_dir, ndays=ndays)

    class Hillslope(ONEDSetup):
        @roger_routine
        def set_settings(self, state):
            s = state.settings
            s.identifier = "ONED"
            s.output_frequency = 86400
            s.nx, s.ny = NX, NY
            s.runlen = 86400 * ndays
            s.nitt_forc = len(forcing["PREC"])
            s.dx = s.dy = 1
            s.x_origin = s.y_origin = 0.0
            s.time_origin = "2019-10-31 23:00:00"     # the example's forcing starts on 1 November 2019
            s.enable_routing_1D = True
            s.enable_lateral_flow = True
            s.enable_groundwater_boundary = False
            s.enable_macropore_lower_boundary_condition = False
            s.enable_adaptive_time_stepping = True

        @roger_routine
        def set_topography(self, state):
            vs = state.variables
            vs.flow_dir_topo = update(vs.flow_dir_topo, at[2:-2, 2:-2], 4)      # towards +y ("south"), oneD.py:131-135
            vs.inner_boundary = update(vs.inner_boundary, at[2:-2, 2:-2], 1)    # the ring of outer_boundary lies in the ghost frame
            vs.maskCatch = update(vs.maskCatch, at[2:-2, 2:-2], 1)

        @roger_routine
        def set_parameters_setup(self, state):
            vs = state.variables
            for name, value in PARAMETERS.items():
                setattr(vs, name, update(getattr(vs, name), at[2:-2, 2:-2], value))
            vs.slope_per = update(vs.slope_per, at[2:-2, 2:-2], vs.slope[2:-2, 2:-2] * 100)

        @roger_routine
        def set_initial_conditions(self, state):
            vs = state.variables   # field capacity, oneD.py:472-483
            for k in ("theta_rz", "theta_ss"):
                setattr(vs, k, update(getattr(vs, k), at[2:-2, 2:-2, : vs.taup1],
                                      (vs.theta_pwp[2:-2, 2:-2] + vs.theta_ufc[2:-2, 2:-2])[:, :, np.newaxis]))

        @roger_routine
        def set_diagnostics(self, state):
            d = state.diagnostics
            for kind, names in (("rate", OUTPUT_RATE), ("collect", OUTPUT_COLLECT)):
                d[kind].output_variables = names
                d[kind].output_frequency = 86400
                d[kind].sampling_frequency = 1
                d[kind].base_output_path = out_dir

    return Hillslope(forcing=forcing, nx=NX, ny=NY, ndays=ndays)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("input_dir", help="directory with PREC.txt, TA.txt, PET.txt")
    ap.add_argument("--days", type=int, default=30)
    ap.add_argument("--out", default="output")
    args = ap.parse_args(argv)
    os.makedirs(args.out, exist_ok=True)
    model = make_model(args.input_dir, args.days, args.out)
    model.setup()
    vs = model.state.variables
    end = args.days * 86400
    while vs.time < end:
        model.run_device(64 if end - vs.time > 65 * 86400 else 1)
    print(f"{vs.itt} steps, {vs.time // 86400} days; ponding at the foot of the slope {float(np.asarray(vs.z0)[2, -3, 1]):.2f} mm; "
          f"output in {args.out}/ONED.rate.nc, ONED.collect.nc")
    return model


if __name__ == "__main__":
    main()
