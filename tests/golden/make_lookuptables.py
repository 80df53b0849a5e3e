#!/usr/bin/env python3
"""Record the reference's look-up tables (model *data*: roger/look_up_tables/*.csv as parsed by
roger/lookuptables.py) into roger_amd/lookuptables.npz.  Build container only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REPO, import_reference  # noqa: E402

if __name__ == "__main__":
    import_reference()
    import roger.lookuptables as lut

    out = os.path.join(REPO, "roger_amd", "lookuptables.npz")
    np.savez_compressed(out, **{k: np.asarray(getattr(lut, k), dtype=np.float64)
                                for k in ("ARR_ILU", "ARR_GC", "ARR_GCM", "ARR_RDLU", "ARR_MLMS", "ARR_IS")})
    print(out, os.path.getsize(out))
