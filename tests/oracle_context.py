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
        self.st.settings.nx, self.st.settings.ny = self.nx, self.ny   # the routing gathers over the (local) grid
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
        self.words = [0, 0, 0, 0]

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
        self.stations = self.weights = None

    def set_forcing_stations(self, F, station_index):
        self.series = {k: np.asarray(v) for k, v in F.items()}
        self.stations = np.asarray(station_index, dtype=np.int64).reshape(-1)
        self.weights = getattr(self, "weights", None)

    def set_forcing_weights(self, prec_weight=None, ta_offset=None, pet_weight=None):
        self.weights = None if prec_weight is None else tuple(np.asarray(a, dtype=np.float64).reshape(-1) for a in (prec_weight, ta_offset, pet_weight))

    def _day_from_series(self, i):
        F = self.series
        st, w = getattr(self, "stations", None), getattr(self, "weights", None)
        if st is not None:
            rows = [np.where(st[:, None] >= 0, np.asarray(F[k])[np.maximum(st, 0), i:i + 144], 0.0) for k in ("PREC", "TA", "PET")]
        elif w is not None:
            rows = [np.broadcast_to(np.asarray(F[k])[i:i + 144], (self.n, 144)) for k in ("PREC", "TA", "PET")]
        else:
            return [F[k][i:i + 144] for k in ("PREC", "TA", "PET")]
        w = w or (np.ones(self.n), np.zeros(self.n), np.ones(self.n))
        return [rows[0] * w[0][:, None], rows[1] + w[1][:, None], rows[2] * w[2][:, None]]

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
            # (several ranks: the word the ranks agreed on in the adaptive time stepping, adaptive_dt_finish)
            w1 = self.__dict__.pop("_agreed_word1", None)
            self._after_adt(monthly=False, core_only=True, word1=self._local_word1() if w1 is None else w1)
            return
        if entry in ("rh_surface_routing", "rh_subsurface_routing"):
            (st.surface_routing if entry == "rh_surface_routing" else st.subsurface_routing)()
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
        self._accumulate()

    # the device-side output accumulators (rh_diag_*), restated: a slot per day of the step's start
    def diag_configure(self, rate=(), collect=(), n_slots=1):
        self._diag = dict(rate=list(rate), collect=list(collect), n_slots=int(n_slots), interval=86400,
                          data={v: np.zeros((int(n_slots), self.n)) for v in list(rate) + list(collect)},
                          steps=np.zeros(int(n_slots), dtype=np.int64), t0=np.full(int(n_slots), -1, dtype=np.int64),
                          t1=np.full(int(n_slots), -1, dtype=np.int64))

    def diag_set_interval(self, seconds):
        self._diag["interval"] = int(seconds)

    def diag_slot_times(self, slot):
        return int(self._diag["t0"][int(slot)]), int(self._diag["t1"][int(slot)])

    def _accumulate(self):
        d = getattr(self, "_diag", None)
        if not d:
            return
        s = self.st.scal
        t0, iv = s.time - s.dt_secs, d["interval"]
        slot, first = (t0 // iv) % d["n_slots"], t0 % iv == 0
        d["steps"][slot] = 1 if first else d["steps"][slot] + 1
        if first:
            d["t0"][slot] = t0
        d["t1"][slot] = s.time
        for v in d["rate"]:
            d["data"][v][slot] = self.st.planes[v] if first else d["data"][v][slot] + self.st.planes[v]
        for v in d["collect"]:
            d["data"][v][slot] = self.st.planes[v]

    def diag_download(self, name, slot):
        return self._diag["data"][name][int(slot)].copy()

    def diag_steps(self, slot):
        return int(self._diag["steps"][int(slot)])

    def diag_upload(self, name, slot, values):
        self._diag["data"][name][int(slot)] = np.asarray(values, dtype=np.float64).reshape(-1)

    def diag_set_slot_state(self, slot, steps, t_start, t_end):
        self._diag["steps"][int(slot)], self._diag["t0"][int(slot)], self._diag["t1"][int(slot)] = int(steps), int(t_start), int(t_end)

    def _hooks(self):
        s, F = self.st.scal, self.series
        if s.time % 86400 == 0 and s.itt_forc + 144 <= len(F["YEAR"]):
            i = s.itt_forc
            s.itt_day = 0
            s.year[1], s.month[1], s.doy[1] = int(F["YEAR"][i]), int(F["MONTH"][i]), int(F["DOY"][i])
            self.set_forcing_day(*self._day_from_series(i))
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

    # one-exchange step (summary protocol of include/roger_hip.h: rh_step_summary / rh_step_finish)
    QB = dict(SWE_NOT_LE0=0, SWE_GT0=1, SWETOP_NOT_LE0=2, SWETOP_GT0=3, RAIN_KEEP=4, SNOWMELT_KEEP=5, P_NOT_LE0=6,
              NOT_PGT0_TALE=7, P_EQ0=8, P_NE0=9)

    def summary_word(self):
        """OR over the local columns of the summary bits (what the fused kernel's waves store at the end of a step)."""
        P, ta_fm = self.st.planes, self.st.settings.ta_fm
        swe, top, prec, ta = P["swe"], P["swe_top"], P["prec"], P["ta"]
        warm = ta > ta_fm
        terms = dict(SWE_NOT_LE0=~(swe <= 0), SWE_GT0=swe > 0, SWETOP_NOT_LE0=~(top <= 0), SWETOP_GT0=top > 0,
                     RAIN_KEEP=(prec > 0) & warm, SNOWMELT_KEEP=((swe > 0) | (top > 0)) & warm, P_NOT_LE0=~(prec <= 0),
                     NOT_PGT0_TALE=~((prec > 0) & (ta <= ta_fm)), P_EQ0=prec == 0, P_NE0=prec != 0)
        return sum(int(np.any(v)) << self.QB[k] for k, v in terms.items())

    def finish_from_summary(self, s_word, monthly=-1):
        """Both predicate words from the (global) summary word, then the rest of the step.  Returns (word0, word1,
        locally evaluated word1) so that single-domain tests can check the derivation against the direct evaluation."""
        L = ob.lib()
        L.oc_adt_pred1.restype = C.c_uint64
        L.oc_adt_select.restype = C.c_uint64
        f = self._forc()
        w0 = (s_word & 0xF) | int(L.oc_adt_pred1(self.st._ptrs, C.c_int64(0), f[0], f[1], f[3], C.byref(self.st.settings)))
        sel = int(L.oc_adt_sel_p(C.c_uint64(w0), C.byref(self.st.scal)))
        w1_local = int(L.oc_adt_select(self.st._ptrs, C.c_int64(self.n), *f, C.byref(self.st.scal),
                                       C.byref(self.st.settings), C.c_uint64(w0)))   # applies the selection to the columns
        b = lambda name: (s_word >> self.QB[name]) & 1  # noqa: E731
        w1 = 0
        if sel >= 0:
            Pv, Tv, ta_fm = float(self.st.planes["prec"][0]), float(self.st.planes["ta"][0]), self.st.settings.ta_fm
            warm = Tv > ta_fm
            w1 |= int((Pv > 0) and warm) << 0
            w1 |= int((b("SWE_GT0") or b("SWETOP_GT0")) and warm) << 1
            w1 |= int(not (Pv <= 0)) << 2
            w1 |= int(not ((Pv > 0) and (Tv <= ta_fm))) << 3
            w1 |= int(Pv == 0) << 6
            w1 |= int(Pv != 0) << 8
        else:
            w1 |= b("RAIN_KEEP") << 0 | b("SNOWMELT_KEEP") << 1 | b("P_NOT_LE0") << 2 | b("NOT_PGT0_TALE") << 3
            w1 |= b("P_EQ0") << 6 | b("P_NE0") << 8
        w1 |= b("SWE_GT0") << 4 | b("SWE_NOT_LE0") << 5 | b("P_NE0") << 7 | b("P_EQ0") << 9
        m = self.monthly if int(monthly) < 0 else bool(monthly)
        L.oc_adt_finish(self.st._ptrs, C.c_int64(self.n), *f, C.byref(self.st.scal), C.byref(self.st.settings),
                        C.c_uint64(w0), C.c_uint64(w1))
        self._after_adt(m, False, w1)
        return w0, w1, w1_local

    def set_time_limit(self, t_end):
        self._t_end = None if t_end is None or int(t_end) < 0 else int(t_end)

    def run_steps(self, nsteps):
        for _ in range(int(nsteps)):
            if getattr(self, "_t_end", None) is not None and int(self.st.scal.time) >= self._t_end:
                return   # rh_set_time_limit: the launches that would follow do nothing
            self._hooks()
            self.step(-1)

    def make_phases(self, one_exchange=True):
        return OraclePhases(self, one_exchange=one_exchange)

    def adaptive_dt_finish(self):
        ob.lib().oc_adt_finish(self.st._ptrs, C.c_int64(self.n), *self._forc(), C.byref(self.st.scal),
                               C.byref(self.st.settings), C.c_uint64(self.words[0]), C.c_uint64(self.words[1]))
        self._agreed_word1 = int(self.words[1])

    # -- routing over several ranks: what rh_surface_routing / rh_subsurface_routing do over RCCL, here over the process group ----
    def comm_init_torch(self, group=None):
        """The edge columns of q_*_out (and, once, of flow direction and mask) go to the x-neighbours between the out and the in
        part of a routing (oracle: oc_route_set_exchange / oc_route_set_halo)."""
        import torch
        import torch.distributed as dist

        rank, world = dist.get_rank(group), dist.get_world_size(group)
        P, nx, ny, L = self.st.planes, self.nx, self.ny, ob.lib()
        self._halo = {"q": [np.zeros(ny), np.zeros(ny)], "fd": [np.zeros(ny, np.int32), np.zeros(ny, np.int32)],
                      "mk": [np.zeros(ny, np.int32), np.zeros(ny, np.int32)], "static": False}

        def swap(edges, dtype):
            """edges = (lo, hi) of this rank -> (halo lo, halo hi) from the neighbours (None where there is none)."""
            ops, got = [], [None, None]
            for side, peer in ((0, rank - 1), (1, rank + 1)):
                if 0 <= peer < world:
                    got[side] = torch.zeros(ny, dtype=dtype)
                    ops.append(dist.P2POp(dist.isend, torch.from_numpy(np.ascontiguousarray(edges[side])).to(dtype), peer, group))
                    ops.append(dist.P2POp(dist.irecv, got[side], peer, group))
            for r in dist.batch_isend_irecv(ops) if ops else []:
                r.wait()
            return got

        def exchange(which):
            H = self._halo
            if not H["static"]:
                fd, mk = P["flow_dir_topo"].reshape(nx, ny), P["maskCatch"].reshape(nx, ny)
                for key, a in (("fd", fd), ("mk", mk)):
                    for side, t in enumerate(swap((a[0], a[-1]), torch.int32)):
                        if t is not None:
                            H[key][side][:] = t.numpy()
                H["static"] = True
            q = P["q_sur_out" if which == 0 else "q_sub_out"].reshape(nx, ny)
            for side, t in enumerate(swap((q[0], q[-1]), torch.float64)):
                if t is None:
                    L.oc_route_set_halo(side, None, None, None)
                else:
                    H["q"][side][:] = t.numpy()
                    L.oc_route_set_halo(side, H["q"][side].ctypes.data_as(C.c_void_p), H["fd"][side].ctypes.data_as(C.c_void_p),
                                        H["mk"][side].ctypes.data_as(C.c_void_p))

        self._exchange_cb = C.CFUNCTYPE(None, C.c_int)(exchange)
        L.oc_route_set_halo.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oc_route_set_exchange(self._exchange_cb)

    def close(self):
        if getattr(self, "_exchange_cb", None) is not None:
            L = ob.lib()
            L.oc_route_set_exchange(None)
            L.oc_route_set_halo.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
            for side in (0, 1):
                L.oc_route_set_halo(side, None, None, None)
            self._exchange_cb = None


class OraclePhases:
    """PhasedStepper backend over an OracleContext, exchanging the predicate words as 64 int32
    0/1 CPU tensors (what the HIP backend does with device tensors)."""

    def __init__(self, ctx, one_exchange=False):
        self.ctx = ctx
        self.one_exchange = one_exchange

    def summary_phase(self):
        self.ctx._hooks()
        self.ctx.words[3] = self.ctx.summary_word()

    def finish_phase(self):
        self.ctx.finish_from_summary(self.ctx.words[3], -1)

    def hooks_phase(self):
        self.ctx._hooks()

    def phase1(self):
        self.ctx.phase1()

    def phase2(self):
        self.ctx.phase2()

    def phase3(self):
        self.ctx.step_phase3(-1)

    def adaptive_dt_finish(self):
        self.ctx.adaptive_dt_finish()

    def predicate_buffer(self, word):
        import torch

        w = self.ctx.words[word]
        return torch.tensor([(w >> b) & 1 for b in range(64)], dtype=torch.int32)

    def load_predicates(self, word, tensor):
        self.ctx.words[word] = sum((1 << b) for b in range(64) if int(tensor[b]))
