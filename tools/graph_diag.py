"""Wall time per step of rh_run_steps with and without replaying 16 captured steps as a hipGraph (an experiment that is
not in the tree any more: the context needs `use_graphs`).  Result on MI355X, no timing events, alternating in one process:
80 x 53 columns 36.3 - 38.1 us/step with graphs, 37.8 - 38.0 plain; 316 x 316: 53.3 - 53.7 vs 53.3 - 53.4; 1000 x 1000:
364.3 - 365.6 vs 364.3 - 365.9.  The step is a chain of two dependent kernels whose launches the host issues in 9 us, far
ahead of the device; a graph removes host work that is not on the critical path.
Usage: python tools/graph_diag.py NX NY STEPS."""
import sys
import time

import torch

sys.path.insert(0, ".")
from roger_amd.forcing import combo_forcing  # noqa: E402
from roger_amd.svat import create_svat, hetero_params  # noqa: E402

nx, ny, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
ctx = create_svat(nx, ny, params=hetero_params(nx * ny, seed=42), device=0)
ctx.set_forcing_series(combo_forcing(ndays=steps // 20 + 40))
ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)
ctx.run_steps(50)
for rep in range(6):
    if hasattr(ctx, "use_graphs"):
        ctx.use_graphs(rep % 2 == 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.run_steps(steps)
    torch.cuda.synchronize()
    print(f"{nx}x{ny} {'graphs' if rep % 2 == 0 else 'plain '}: {(time.perf_counter() - t0) / steps * 1e6:.1f} us/step", flush=True)
s = ctx.get_scalars()
print("itt", s.itt, "time", s.time, "sanity", s.sanity_ok)
ctx.close()
