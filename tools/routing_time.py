#!/usr/bin/env python3
"""Wall-clock of the routed step (settings.enable_routing_1D, rh_run_steps on a routing context) next to the fused oneD step on the
same grid and forcing.   python tools/routing_time.py [nx ny] [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from roger_amd.forcing import combo_forcing  # noqa: E402
from roger_amd.svat import create_svat  # noqa: E402

nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000, 1000)
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
oned = dict(z_soil=1000.0, lmpv=600.0, slope=0.05, slope_per=5, dmph=50.0)
forcing = combo_forcing(ndays=max(30, steps + 15))
for routed in (False, True):
    extra = dict(enable_routing_1D=1, dx=5.0, dy=5.0) if routed else {}
    params = dict(oned, flow_dir_topo=4, k_st=15.0) if routed else oned   # every cell drains towards +y
    ctx = create_svat(nx, ny, params=params, lateral=True, **extra)
    ctx.set_forcing_series(forcing)
    ctx.run_steps(10)
    ctx.sync()
    t0 = time.perf_counter()
    ctx.run_steps(steps)
    ctx.sync()
    dt = time.perf_counter() - t0
    s = ctx.get_scalars()
    print(f"{'routed' if routed else 'fused '} oneD step, {nx} x {ny}: {dt / steps * 1e3:.3f} ms per step, {nx * ny * steps / dt:.3e} cell-timesteps/s "
          f"(itt {s.itt}, z0 max {np.nanmax(ctx.download('z0')):.2f} mm)")
    ctx.close()
