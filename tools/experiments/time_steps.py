"""Wall time per step of rh_run_steps for timing-only library variants (builds whose results are not valid -- the tail or the completion
counting compiled out -- and which therefore cannot pass bench.py's bookkeeping checks).
    ROGER_HIP_LIB=roger_amd/variants/libroger_hip_<v>.so python tools/experiments/time_steps.py nx ny steps [hetero]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from roger_amd.forcing import combo_forcing
from roger_amd.svat import create_svat, hetero_params

nx, ny, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
params = hetero_params(nx * ny, seed=42) if len(sys.argv) > 4 else None
ctx = create_svat(nx, ny, params=params, device=0)
ctx.set_forcing_series(combo_forcing(ndays=max(30, 2 * steps + 150)))
ctx.set_stream(torch.cuda.current_stream(0).cuda_stream)
ctx.run_steps(120)
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    ctx.run_steps(steps)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / steps * 1e3)
print(os.environ.get("ROGER_HIP_LIB", "product"), f"{nx}x{ny}: {best:.4f} ms per step (best of 3 x {steps})")
