"""Land-use look-up tables used by the surface/soil parameter kernels.

Same names as the reference module (`roger/lookuptables.py`: ARR_ILU, ARR_GC, ARR_GCM,
ARR_RDLU), so setup scripts can keep `import ...lookuptables as lut; vs.lut_ilu = lut.ARR_ILU`.
The numbers are model *data* (the reference's `roger/look_up_tables/*.csv` as parsed by its
`lookuptables.py`), stored here as `lookuptables.npz`; tests/golden/make_lookuptables.py records them
from the reference's `roger.lookuptables` module.
"""
import os

import numpy as _np

_d = _np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lookuptables.npz"))
ARR_ILU = _d["ARR_ILU"]    # (25, 13) land use x month: interception storage
ARR_GC = _d["ARR_GC"]      # (25, 13) land use x month: ground cover
ARR_GCM = _d["ARR_GCM"]    # (25, 2)  land use: maximum ground cover
ARR_RDLU = _d["ARR_RDLU"]  # (25, 7)  land use: rooting depth
ARR_MLMS = _d["ARR_MLMS"]  # (10000, 9) slope (%): horizontal macropore flow velocities of layers 8..1 (m/h)
ARR_IS = _d["ARR_IS"]      # (101, 2) sealing: interception storage
