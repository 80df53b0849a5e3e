"""Speed level of the fused kernel against the arena's address: contexts alive at once with different paddings in front
(torch allocations) and different offsets inside their own allocation (RH_ARENA_OFFSET_KB).  Diagnostics only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from roger_amd.forcing import combo_forcing  # noqa: E402
from roger_amd.svat import create_svat  # noqa: E402

device = torch.device("cuda", 0)
torch.cuda.set_device(0)
forcing = combo_forcing(ndays=60)
rows = []
keep = []
spec = [(0, 0), (0, 64), (0, 1024), (0, 4096), (0, 65536), (0, 1048576), (389, 0), (777, 0), (1231, 0), (1231, 1024), (2048, 0), (3000, 0)]
for pad_mb, off_kb in spec:
    if pad_mb:
        keep.append(torch.empty(pad_mb * 1024 * 1024, dtype=torch.uint8, device=device))
    os.environ["RH_ARENA_OFFSET_KB"] = str(off_kb)
    ctx = create_svat(1000, 1000, device=0)
    ctx.set_forcing_series(forcing)
    ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)
    ctx.run_steps(10)
    ctx.enable_timing(True)
    ctx.run_steps(60)
    ms, _ = ctx.timing_detail()
    ctx.enable_timing(False)
    first = next(i for i, (_, is_int) in enumerate(ctx.planes) if not is_int)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        ctx.calibrate_copy(first, first + 100, 96)
    e0.record()
    for _ in range(10):
        ctx.calibrate_copy(first, first + 100, 96)
    e1.record()
    torch.cuda.synchronize()
    copy_ms = e0.elapsed_time(e1) / 10
    base = ctx._lib.rh_plane_device_ptr(ctx._h, 0)
    print(f"pad {pad_mb:5d} MB offset {off_kb:8d} KB  base {base:#016x}  mod 1GiB {base % (1 << 30) >> 20:5d} MiB  mod 2MiB {base % (2 << 20) >> 10:5d} KiB  "
          f"kernel p50 {np.percentile(ms, 50):.3f} ms  copy(96 planes) {copy_ms:.3f} ms", flush=True)
    keep.append(ctx)
