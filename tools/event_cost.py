"""Wall time per step of rh_run_steps with and without the HIP-event pair around the fused kernel.  Diagnostics only."""
import sys
import time

import torch

sys.path.insert(0, ".")
from roger_amd.forcing import combo_forcing  # noqa: E402
from roger_amd.svat import create_svat  # noqa: E402

device = torch.device("cuda", 0)
torch.cuda.set_device(0)
ctx = create_svat(1000, 1000, device=0)
ctx.set_forcing_series(combo_forcing(ndays=60))
ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)
ctx.run_steps(20)
for rep in range(3):
    for timing in (False, True):
        ctx.enable_timing(timing)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.run_steps(200)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        extra = ""
        if timing:
            ms, n = ctx.timing_summary()
            extra = f", kernel {ms / n:.4f} ms"
        print(f"timing={timing}: {dt / 200 * 1e3:.4f} ms/step{extra}", flush=True)
ctx.close()
