"""Placement probing at work: four contexts alive at once (as tools/placement_diag2.py), each reporting what rh_create
probed and the level the fused kernel then runs at.  Diagnostics only."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from roger_amd.forcing import combo_forcing  # noqa: E402
from roger_amd.svat import create_svat  # noqa: E402

device = torch.device("cuda", 0)
torch.cuda.set_device(0)
forcing = combo_forcing(ndays=60)
keep = []
for k in range(4):
    keep.append(torch.empty((k * 389 + 64) * 1024 * 1024, dtype=torch.uint8, device=device))
    ctx = create_svat(1000, 1000, device=0)
    ctx.set_forcing_series(forcing)
    ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)
    ctx.run_steps(10)
    ctx.enable_timing(True)
    ctx.run_steps(60)
    ms, _ = ctx.timing_detail()
    ctx.enable_timing(False)
    print(f"ctx {k}: probes {[round(v, 3) for v in ctx.placement_report()]} -> kernel p50 {np.percentile(ms, 50):.3f} ms", flush=True)
    keep.append(ctx)
