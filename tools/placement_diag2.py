"""Three SVAT contexts alive at once, timed in turn for several rounds: if a context keeps its speed level while the levels
differ between contexts, the level belongs to where its memory landed, not to the clocks.  Diagnostics only."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from roger_amd.forcing import combo_forcing  # noqa: E402
from roger_amd.svat import create_svat  # noqa: E402

device = torch.device("cuda", 0)
torch.cuda.set_device(0)
forcing = combo_forcing(ndays=60)
ctxs = []
pads = []
for k in range(4):
    pads.append(torch.empty((k * 389 + 64) * 1024 * 1024, dtype=torch.uint8, device=device))
    ctx = create_svat(1000, 1000, device=0)
    ctx.set_forcing_series(forcing)
    ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)
    ctx.run_steps(10)
    ctxs.append(ctx)
for rnd in range(4):
    row = []
    for ctx in ctxs:
        ctx.enable_timing(True)
        ctx.run_steps(60)
        ms, _ = ctx.timing_detail()
        ctx.enable_timing(False)
        row.append(float(np.percentile(ms, 50)))
    print("round", rnd, " ".join(f"{v:.3f}" for v in row), flush=True)
