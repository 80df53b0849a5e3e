"""CPU test double of `roger_amd._native.Context`, backed by the oracle.

Lets the `-m "not gpu"` suite exercise the host package (RogerSetup hook order, variable
synchronisation, decomposition, the phased multi-rank step) without a GPU.  Test
infrastructure only: nothing in roger_amd imports it, and it is never a fallback of the product.
"""
import ctypes as C

import numpy as np

import oracle_binding as ob
from roger_amd import _native as N

_SC = ("itt", "time", "dt_secs", "itt_day", "itt_forc", "time_event0", "event_id_counter", "dt", "sanity_ok")
_SC2 = ("event_id", "year", "month", "doy")


class OracleContext:
    def __init__(self, nx, ny, device=0, **settings):
        self.nx, self.ny, self.n = int(nx), int(ny), int(nx) * int(ny)
        self.st = ob.OracleState(self.n)
        for k, v in settings.items():
            setattr(self.st.settings, k, v)
        self.mlms = None
        self.planes = list(zip(self.st.names, self.st.is_int))
        self.index = {nm: i for i, nm in enumerate(self.st.names)}
        P = self.st.planes
        P["maskCatch"][:] = 1
        for nm, v in (("ta", 15.0), ("ta_m1", 15.0), ("z_gw", 1000.0), ("z_gw_m1", 1000.0), ("c_int", 1.0), ("c_root", 1.0)):
            P[nm][:] = v
        s = self.st.scal
        s.dt, s.dt_secs, s.event_id_counter, s.sanity_ok = 1.0, 3600, 1, 1
        for k, v in (("year", 1900), ("month", 1), ("doy", 1)):
            getattr(s, k)[0] = getattr(s, k)[1] = v
        self.day = None
        self.series = None
        self.monthly = False
        self.words = [0, 0]

    def close(self):
        pass

    def sync(self):
        pass

    def set_stream(self, handle):
        pass

    # planes
    def dtype_of(self, name):
        return self.st.planes[name].dtype.type

    def upload(self, name, arr):
        a = np.asarray(arr).reshape(-1)
        assert a.size == self.n
        self.st.planes[name][:] = a.astype(self.st.planes[name].dtype)

    def download(self, name):
        return self.st.planes[name].copy()

    # scalars
    def set_scalars(self, s):
        for k in _SC:
            setattr(self.st.scal, k, getattr(s, k))
        for k in _SC2:
            getattr(self.st.scal, k)[0], getattr(self.st.scal, k)[1] = getattr(s, k)[0], getattr(s, k)[1]

    def get_scalars(self):
        s = N.RhScalars()
        for k in _SC:
            setattr(s, k, getattr(self.st.scal, k))
        for k in _SC2:
            getattr(s, k)[0], getattr(s, k)[1] = getattr(self.st.scal, k)[0], getattr(self.st.scal, k)[1]
        return s

    def set_luts(self, ilu, gc, gcm, rdlu):
        self.st.set_luts(ilu, gc, gcm, rdlu)

    def set_lut_mlms(self, mlms):
        self.mlms = np.ascontiguousarray(mlms, dtype=np.float64)

    def set_forcing_day(self, prec_day, ta_day, pet_day):
        self.day = [np.ascontiguousarray(a, dtype=np.float64) for a in (prec_day, ta_day, pet_day)]

    def set_forcing_series(self, F):
        self.series = {k: np.asarray(v) for k, v in F.items()}

    # routines
    def call(self, entry):
        st = self.st
        if entry == "rh_hooks_phase":
            return self._hooks()
        if entry == "rh_step_phase1":
            return self.phase1()
        if entry == "rh_step_phase2":
            return self.phase2()
        if entry == "rh_adaptive_dt":
            st.adaptive_dt(*self.day)
            return
        if entry == "rh_step_core":
            self._after_adt(monthly=False, core_only=True, word1=self._local_word1())
            return
        if entry == "rh_params_lateral":
            st.params_lateral(self.mlms)
            return
        {"rh_topo": st.topo, "rh_params_surface": st.params_surface, "rh_params_soil": st.params_soil,
         "rh_initial_conditions": st.initial_conditions, "rh_interception": st.interception,
         "rh_evapotranspiration": st.evapotranspiration, "rh_snow": st.snow, "rh_infiltration": st.infiltration,
         "rh_subsurface_runoff": st.subsurface_runoff, "rh_capillary_rise": st.capillary_rise,
         "rh_storage": st.storage, "rh_num_error": st.num_error, "rh_after_timestep": st.after_timestep,
         "rh_sync": lambda: None}[entry]()

    def _forc(self):
        return self.st._forc(*self.day)

    def _local_word1(self):
        P = self.st.planes
        w = 0
        w |= int(np.any(P["prec"] == 0)) << 6
        w |= int(np.any(P["prec_m1"] != 0)) << 7
        w |= int(np.any(P["prec"] != 0)) << 8
        w |= int(np.any(P["prec_m1"] == 0)) << 9
        return w

    def _after_adt(self, monthly, core_only, word1):
        L = ob.lib()
        L.oc_step_after_adt.restype = C.c_int
        L.oc_step_after_adt(self.st._ptrs, C.c_int64(self.n), C.byref(self.st.scal), C.byref(self.st.settings),
                            C.c_int(int(monthly)), C.c_int(int(core_only)), C.c_uint64(word1), *self.st._lut_args())

    def _hooks(self):
        s, F = self.st.scal, self.series
        if s.time % 86400 == 0 and s.itt_forc + 144 <= len(F["PREC"]):
            i = s.itt_forc
            s.itt_day = 0
            s.year[1], s.month[1], s.doy[1] = int(F["YEAR"][i]), int(F["MONTH"][i]), int(F["DOY"][i])
            self.set_forcing_day(F["PREC"][i:i + 144], F["TA"][i:i + 144], F["PET"][i:i + 144])
            s.itt_forc = i + 144
        self.monthly = (s.month[1] != s.month[0]) and (s.itt > 1)

    # three-phase step (PhasedStepper protocol)
    def phase1(self):
        L = ob.lib()
        L.oc_adt_pred1.restype = C.c_uint64
        f = self._forc()
        self.words[0] = int(L.oc_adt_pred1(self.st._ptrs, C.c_int64(self.n), f[0], f[1], f[3], C.byref(self.st.settings)))

    def phase2(self):
        L = ob.lib()
        L.oc_adt_select.restype = C.c_uint64
        self.words[1] = int(L.oc_adt_select(self.st._ptrs, C.c_int64(self.n), *self._forc(), C.byref(self.st.scal),
                                            C.byref(self.st.settings), C.c_uint64(self.words[0])))

    def step_phase3(self, monthly=False):
        m = self.monthly if int(monthly) < 0 else bool(monthly)
        ob.lib().oc_adt_finish(self.st._ptrs, C.c_int64(self.n), *self._forc(), C.byref(self.st.scal),
                               C.byref(self.st.settings), C.c_uint64(self.words[0]), C.c_uint64(self.words[1]))
        self._after_adt(m, False, self.words[1])

    def step(self, monthly=False):
        self.phase1()
        self.phase2()
        self.step_phase3(monthly)

    def run_steps(self, nsteps):
        for _ in range(int(nsteps)):
            self._hooks()
            self.step(-1)


class OraclePhases:
    """PhasedStepper backend over an OracleContext, exchanging the predicate words as 64 int32
    0/1 CPU tensors (what the HIP backend does with device tensors)."""

    def __init__(self, ctx):
        self.ctx = ctx

    def hooks_phase(self):
        self.ctx._hooks()

    def phase1(self):
        self.ctx.phase1()

    def phase2(self):
        self.ctx.phase2()

    def phase3(self):
        self.ctx.step_phase3(-1)

    def predicate_buffer(self, word):
        import torch

        w = self.ctx.words[word]
        return torch.tensor([(w >> b) & 1 for b in range(64)], dtype=torch.int32)

    def load_predicates(self, word, tensor):
        self.ctx.words[word] = sum((1 << b) for b in range(64) if int(tensor[b]))
