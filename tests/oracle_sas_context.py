"""Oracle-backed stand-in for `roger_amd._native.SasContext` -- test infrastructure for the CPU-only suite.

Same surface (names, shape, dtype, upload, download, step, sync, close) with oracle/sas_oracle.c doing the
arithmetic, so that the host package's offline-transport flow (RogerSetup.setup/step, lazy synchronisation of
RogerVariables, core.transport.calculate_storage_selection) can be exercised without a GPU.  Never imported by
roger_amd/."""
import numpy as np

import sas_binding as sb
from roger_amd._native import DAILY_INPUTS, NativeError


class OracleSasContext:
    def __init__(self, n_cells, ages, substeps=1, device=0, forcing_days=1, age_statistics=False,
                 keep_distributions=False, tracer="oxygen18", solver="deterministic", **settings):
        assert forcing_days == 1
        self.n, self.ages, self.substeps, self.forcing_days = int(n_cells), int(ages), int(substeps), 1
        self.tracer = tracer
        self.solver = solver
        self.st = sb.SasState(self.n, self.ages, self.substeps, age_statistics, tracer=tracer, solver=solver)
        self.keep = keep_distributions
        self.names = list(self._arrays())

    def _arrays(self):
        st = self.st
        out = {"maskCatch": st.maskCatch}
        out.update(st.state)
        out.update(st.inp)
        out.update({f"sas_params_{f}": a for f, a in st.sas.items()})
        out.update(st.S_init)
        out.update(st.out)
        if st.anion:
            out.update(st.par)
        return out

    def shape(self, name):
        a = self._arrays()[name]
        return (1, self.n) if name in DAILY_INPUTS else a.shape

    def dtype(self, name):
        return np.int32 if name in ("maskCatch", "lu_id") else np.float64

    def upload(self, name, host):
        a = np.asarray(host)
        if a.shape != self.shape(name):
            raise ValueError(f"{name}: shape {a.shape}, expected {self.shape(name)}")
        self._arrays()[name][...] = a[0] if name in DAILY_INPUTS else a

    def download(self, name):
        if name not in self._arrays():
            raise NativeError(f"array {name} is not held by this context")
        a = self._arrays()[name].copy()
        return a[None, :] if name in DAILY_INPUTS else a

    def step(self, day):
        self.st.step_oracle()

    def stages(self, day, mask):
        if mask == 511:
            self.st.step_oracle()
        elif mask == 512:
            self.st.rescale_oracle()
        else:
            raise NotImplementedError("the oracle stand-in runs whole days and the rescaling only")

    def sync(self):
        pass

    def close(self):
        pass
