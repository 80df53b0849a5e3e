"""How many candidate placements are fast on this box?  One context with many probes; prints the sorted probe times."""
import os
import sys

os.environ.setdefault("RH_PLACEMENT_PROBES", "24")
sys.path.insert(0, ".")
from roger_amd.svat import create_svat  # noqa: E402

ctx = create_svat(1000, 1000, device=0)
rep = ctx.placement_report()
print("probes (ms):", " ".join(f"{v:.3f}" for v in rep))
print("sorted     :", " ".join(f"{v:.3f}" for v in sorted(rep)))
ctx.close()
