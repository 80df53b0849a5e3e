"""Does the fused kernel's speed depend on where the arena lands in memory?  Several contexts in one process with dummy
allocations of different sizes in between; prints the arena's base address and the median kernel time.  Diagnostics only."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from roger_amd.forcing import combo_forcing  # noqa: E402
from roger_amd.svat import create_svat  # noqa: E402

device = torch.device("cuda", 0)
torch.cuda.set_device(0)
forcing = combo_forcing(ndays=30)
keep = []
for k, pad_mb in enumerate([0, 0, 37, 301, 1024, 5, 2048, 777]):
    if pad_mb:
        keep.append(torch.empty(pad_mb * 1024 * 1024, dtype=torch.uint8, device=device))
    ctx = create_svat(1000, 1000, device=0)
    ctx.set_forcing_series(forcing)
    ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)
    ctx.run_steps(10)
    ctx.enable_timing(True)
    ctx.run_steps(100)
    ms, dts = ctx.timing_detail()
    ctx.enable_timing(False)
    base = ctx._lib.rh_plane_device_ptr(ctx._h, 0)
    print(f"ctx {k}: pad {pad_mb:5d} MB, plane 0 at {base:#x} (mod 2 MiB = {base % (2 << 20):#x}), kernel p50 {np.percentile(ms, 50):.3f} ms", flush=True)
    ctx.close()
    if k % 2 == 1:
        keep.clear()
        torch.cuda.empty_cache()
