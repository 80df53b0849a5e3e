"""Per-launch kernel time of the fused step, direct (rh_run_steps) against the multi-GPU orchestration (PhasedStepper) on
one GPU: percentiles of the HIP-event durations and host time per step.  Diagnostics only."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from roger_amd.distributed import HipPhases, PhasedStepper  # noqa: E402
from roger_amd.forcing import combo_forcing  # noqa: E402
from roger_amd.svat import create_svat  # noqa: E402

device = torch.device("cuda", 0)
torch.cuda.set_device(0)
for mode in sys.argv[1:] or ["direct", "phased", "phased_noexchange"]:
    ctx = create_svat(1000, 1000, device=0)
    ctx.set_forcing_series(combo_forcing(ndays=30))
    if mode == "direct":
        ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)
        run = ctx.run_steps
    else:
        run = PhasedStepper(HipPhases(ctx, device), always_exchange=(mode == "phased")).run
    run(10)
    ctx.enable_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(200)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    ms, dts = ctx.timing_detail()
    print(f"{mode:18s} host-issue {t_host / 200 * 1e3:.3f} ms/step, wall {t_all / 200 * 1e3:.3f} ms/step; kernel ms "
          f"p10 {np.percentile(ms, 10):.3f} p50 {np.percentile(ms, 50):.3f} p90 {np.percentile(ms, 90):.3f} max {ms.max():.3f}")
    ctx.close()
