"""`RogerSetup`: the driver of a model run, mirroring roger/roger.py for the SVAT path.

Same abstract hooks, same call order in `setup()` (roger.py:258-336) and `step()`
(roger.py:355-489).  The process routines between the hooks are native:

    read_data, set_boundary_conditions, set_forcing          user hooks (host)
    adaptive time stepping                                   rh_adaptive_dt
    set_parameters                                           user hook (host); its
                                                             calc_parameters_surface_kernel is native
    interception ... numerics, itt/time                      rh_step_core (one fused kernel)
    after_timestep                                           user hook; its after_timestep_kernel
                                                             is native

`run()` uses that hook-preserving sequence.  `run_device(nsteps)` is the fast path for setups
whose hooks are the benchmark's (forcing series sliced at midnight, monthly surface parameters):
it uploads vs.PREC/TA/PET/YEAR/MONTH/DOY once and advances with rh_run_steps, no host round trip.
"""
import abc
import os

from . import diagnostics, distributed, logger, restart, runtime_settings as rs, runtime_state as rst
from . import settings as settings_mod
from .routines import is_roger_routine, roger_routine, run_native
from .state import RogerState
from .timer import Timer  # noqa: F401


class RogerSetup(metaclass=abc.ABCMeta):
    """Main class for roger setups on the hip backend (roger/roger.py:17-45)."""

    def __init__(self, override=None):
        self.override_settings = override or {}
        from . import core  # noqa: F401  (locks the runtime settings, roger/roger.py:40)

        self.state = RogerState()
        self._setup_done = False

    # -- the abstract hooks, roger/roger.py:47-252 -------------------------------------------
    @abc.abstractmethod
    def set_settings(self, state):
        pass

    @abc.abstractmethod
    def set_grid(self, state):
        pass

    @abc.abstractmethod
    def set_topography(self, state):
        pass

    @abc.abstractmethod
    def set_look_up_tables(self, state):
        pass

    @abc.abstractmethod
    def set_parameters_setup(self, state):
        pass

    @abc.abstractmethod
    def set_parameters(self, state):
        pass

    @abc.abstractmethod
    def set_initial_conditions_setup(self, state):
        pass

    @abc.abstractmethod
    def set_initial_conditions(self, state):
        pass

    @abc.abstractmethod
    def set_boundary_conditions_setup(self, state):
        pass

    @abc.abstractmethod
    def set_boundary_conditions(self, state):
        pass

    @abc.abstractmethod
    def set_forcing_setup(self, state):
        pass

    @abc.abstractmethod
    def set_forcing(self, state):
        pass

    @abc.abstractmethod
    def set_diagnostics(self, state):
        pass

    @abc.abstractmethod
    def after_timestep(self, state):
        pass

    def read_data(self, state):
        pass

    def _ensure_setup_done(self):
        if not self._setup_done:
            raise RuntimeError("setup() method has to be called before running the model")

    # -- setup, roger/roger.py:258-336 ----------------------------------------------------------
    def setup(self):
        from .core import soil, surface

        for f in (self.set_parameters_setup, self.set_grid, self.set_topography, self.set_initial_conditions_setup,
                  self.set_initial_conditions, self.set_boundary_conditions_setup, self.set_boundary_conditions,
                  self.set_diagnostics, self.set_forcing_setup, self.after_timestep):
            if not is_roger_routine(f):
                raise RuntimeError(
                    f"{f.__name__} method is not a roger routine. Please make sure to decorate it "
                    "with @roger_routine and try again.")
        state = self.state
        with state.timers["setup"]:
            with state.settings.unlock():
                self.set_settings(state)
                for setting, value in self.override_settings.items():
                    setattr(state.settings, setting, value)
            settings_mod.check_setting_conflicts(state.settings)
            # against the REAL size of the process group (roger/roger.py:292, distributed.py:121-138): a run started on N ranks
            # without num_proc=(N, 1) must fail here instead of stepping N uncoupled copies
            distributed.validate_decomposition(state.settings.nx, state.settings.ny, rs.num_proc, rst.proc_num)
            if rst.proc_num > 1 and rs.num_proc[1] != 1:
                raise NotImplementedError("the hip backend splits the grid along x only: num_proc = (N, 1) (BASELINE.json north_star)")
            state.initialize_variables()
            offline = state.settings.enable_offline_transport
            if rst.proc_num > 1 and state.settings.enable_routing_1D and not offline:
                # routed water crosses the rank boundaries: the edge columns travel over the context's communicator inside
                # rh_step_core / rh_step_routed (the reference never exchanges them, core/utilities.py:79 is not on this path)
                state.backend_context.comm_init_torch()
                self._comm_ready = True
            self.set_grid(state)
            self.set_topography(state)
            self.set_look_up_tables(state)
            if not offline:
                self._upload_luts()
            self.set_parameters_setup(state)
            if not offline:   # the parameter / initial-condition kernels are skipped for offline transport:
                surface.calculate_parameters(state)   # roger/core/surface.py:391,425, soil.py:731,1002
                soil.calculate_parameters(state)
            self.set_initial_conditions_setup(state)
            self.set_initial_conditions(state)
            if not offline:
                surface.calculate_initial_conditions(state)
                soil.calculate_initial_conditions(state)
            state.diagnostics.update(diagnostics.create_default_diagnostics(state))   # roger/roger.py:296
            self.set_diagnostics(state)
            diagnostics.initialize(state)
            self.set_boundary_conditions_setup(state)
            self.set_boundary_conditions(state)
            self.set_forcing_setup(state)
            restart.read_restart(state)   # roger/roger.py:324-326
        self._setup_done = True
        if not state.settings.enable_offline_transport:   # roger/roger.py:324-327
            with state.settings.unlock():
                state.settings.warmup_done = True

    def warmup(self, repeat=1):
        """roger/roger.py:491-521: for offline transport `repeat` whole runs, each followed by soil.rescale_SA, then
        itt = time = 0; warmup_done is set either way."""
        from .core import soil

        if self.state.settings.enable_offline_transport:
            with self.state.timers["warmup"]:
                for _ in range(repeat):
                    self.run()
                    soil.rescale_SA(self.state)
                with self.state.variables.unlock():
                    self.state.variables.itt = 0
                    self.state.variables.time = 0
        with self.state.settings.unlock():
            self.state.settings.warmup_done = True
        diagnostics.output_transport(self.state)   # initial values after the warm-up, roger/roger.py:515-521

    def _upload_luts(self):
        vs = self.state.variables
        self.state.backend_context.set_luts(vs.lut_ilu, vs.lut_gc, vs.lut_gcm, vs.lut_rdlu)
        if self.state.settings.enable_lateral_flow:
            self.state.backend_context.set_lut_mlms(vs.lut_mlms)

    # -- one time step, roger/roger.py:355-489 ------------------------------------------------------
    @roger_routine
    def step(self, state):
        self._ensure_setup_done()
        if state.settings.enable_offline_transport:
            return self._step_offline_transport(state)
        if state.settings.restart_frequency > 0:
            with state.timers["diagnostics"]:
                restart.write_restart(state)   # roger/roger.py:385-386
        with state.timers["main"]:
            with state.timers["read data"]:
                self.read_data(state)
            with state.timers["boundary conditions"]:
                self.set_boundary_conditions(state)
            with state.timers["forcing"]:
                self.set_forcing(state)
            if self._fused_host_step_possible():
                # The script brought hooks of its own for what comes BEFORE the physics (read_data, set_boundary_conditions, set_forcing)
                # and left set_parameters and after_timestep to the model class: the rest of the step -- adaptive time step, the monthly
                # surface parameters, the processes, the rotation -- is the fused kernel's, one native call instead of three and
                # 1 768 instead of 3 300 B per column (rh_svat_step; the month change is the stock hook's own test, models/svat.py).
                with state.timers["processes"]:
                    vs = state.variables
                    monthly = bool((vs.month[vs.tau] != vs.month[vs.taum1]) & (vs.itt > 1))
                    vs.flush_to_device()
                    if hasattr(state.backend_context, "step_scalars"):
                        vs.mark_device_newer(None, scalars=state.backend_context.step_scalars(monthly))
                    else:   # (the oracle double of the CPU tests)
                        state.backend_context.step(monthly)
                        vs.mark_device_newer(None)
                return self._end_of_step(state)
            with state.timers["adaptive time-stepping"]:
                if rst.proc_num > 1:
                    # dt is ONE scalar for the whole domain: the ranks agree on the predicates before it is derived
                    # (adaptive_time_stepping_dist_safe.py:6-26 does it through rank 0)
                    state.variables.flush_to_device()
                    self._stepper(one_exchange=False).adaptive_dt()
                    state.variables.mark_device_newer(("prec", "ta", "pet", "pet_res"))
                else:
                    run_native(state, "rh_adaptive_dt", ("prec", "ta", "pet", "pet_res"))
            with state.timers["time-variant parameters"]:
                self.set_parameters(state)
            with state.timers["processes"]:
                run_native(state, "rh_step_core")
        self.after_timestep(state)
        self._end_of_step(state)

    @staticmethod
    def _end_of_step(state):
        if getattr(state, "_diag_active", None):   # roger/roger.py:458-465: output at the end of the time step
            diagnostics.output(state)
        if rs.profile_mode:
            state.backend_context.sync()
            logger.info(" Time step took {:.2f}s".format(state.timers["main"].last_time))

    def _step_offline_transport(self, state):
        """roger/roger.py:466-485: the offline-transport branch of step()."""
        from .core import transport

        vs = state.variables
        with state.timers["main"]:
            with vs.unlock():
                vs.itt = vs.itt + 1   # skip first iteration which contains initial values
                if state.settings.sas_solver == "deterministic":
                    vs.time = vs.time + vs.dt_secs
            with state.timers["main transport"]:
                with state.timers["read data"]:
                    self.read_data(state)
                with state.timers["boundary conditions"]:
                    self.set_boundary_conditions(state)
                with state.timers["forcing"]:
                    self.set_forcing(state)
                with state.timers["time-variant parameters"]:
                    self.set_parameters(state)
                with state.timers["StorAge selection"]:
                    transport.calculate_storage_selection(state)
                diagnostics.output_transport(state)    # write_output, roger/core/transport.py:3399-3418
        self.after_timestep(state)
        if rs.profile_mode:
            state.sas_context.sync()
            logger.info(" Time step took {:.2f}s".format(state.timers["main"].last_time))

    # the per-step user hooks; when all of them do what the device-side control part does itself (roger_hip.hip ctrl_wave), run()
    # needs no host code between two steps.  Decided per hook by BEHAVIOUR (roger_amd/hooks.py): every script the reference ships
    # defines these hooks itself (benchmarks/SVAT_benchmark.py:105-110, 152-181), so inheritance says nothing.
    STEP_HOOKS = ("read_data", "set_boundary_conditions", "set_forcing", "set_parameters", "after_timestep")
    recognise_hooks = True   # a script sets this to False to keep every hook of its own on the host, unprobed

    def hook_classes(self):
        """{hook: True if the device performs it} -- the hooks of the ready-made model classes by their mark, a script's own by
        probing them once against the recording state (after setup(): the probes look at the forcing series and weights)."""
        self._ensure_setup_done()
        if getattr(self, "_hook_classes", None) is None:
            from . import hooks

            self._hook_classes = hooks.classify(self, self.STEP_HOOKS)
            self.state._stock_set_forcing = bool(self._hook_classes["set_forcing"])   # (restart.collect: the day arrays at midnight)
        return self._hook_classes

    def _fused_host_step_possible(self):
        """step(): the hooks BEHIND set_forcing (set_parameters, after_timestep) are the device's own -- the model class's, or a script's
        that do the same --, one rank, no routing: the physics of the step is one native call (rh_svat_step)."""
        settings = self.state.settings
        if rst.proc_num > 1 or settings.enable_routing_1D or settings.enable_offline_transport:
            return False
        # (RH_STEP_BY_ROUTINE=1: the three-call step of rounds 1 - 3, for A/B and for the tests of that path)
        if not hasattr(self.state.backend_context, "step") or os.environ.get("RH_STEP_BY_ROUTINE"):
            return False
        classes = self.hook_classes()
        return classes["set_parameters"] and classes["after_timestep"]

    def device_run_possible(self):
        """True if `run()` may advance on the device without returning to the host between steps: the setup script left the
        per-step hooks to the model class (SVATSetup / ONEDSetup: forcing series sliced at midnight, monthly surface parameters,
        tau -> taum1 rotation), and nothing was asked for that needs the host after every step."""
        settings = self.state.settings
        if settings.enable_offline_transport or rs.profile_mode or settings.restart_frequency > 0:
            return False
        if not hasattr(self.state.backend_context, "run_steps") or os.environ.get("RH_STEP_BY_ROUTINE"):
            return False
        return all(self.hook_classes().values())

    def _run_on_device(self, start_time, runlen):
        """`while vs.time - start_time < runlen: step()` (roger/roger.py:548-556) without the host in the loop.  The step length is
        decided on the device, so the number of steps is not known beforehand.  The device is given the end of the run
        (rh_set_time_limit): the control part of a step finds the run over and the launches behind it do nothing, so rounds of steps
        may be enqueued generously -- the first one ceil(remaining / day) steps (a step covers at most a day), the following ones what
        the mean step length so far suggests plus a margin -- with the time read back after each round: a handful of
        synchronisations per run instead of three native calls per step.  Output intervals are fetched before their slots on the
        device are reused.  Without a time limit (per-cell forcing, routing: their control parts do not observe it) every round is
        ceil(remaining / day) steps, which can never overshoot."""
        state = self.state
        vs = state.variables
        ctx = state.backend_context
        if not getattr(self, "_device_hooks", False):
            self.enable_device_hooks()
        t_stop = start_time + runlen
        slots = iv = None
        if getattr(state, "_diag_active", None) and not getattr(state, "_diag_transport", False):
            slots, iv = state._diag_slots, state._diag_interval   # output intervals resident on the device; their length
        limit = hasattr(ctx, "set_time_limit") and not getattr(self, "_per_cell_forcing", False) and not state.settings.enable_routing_1D
        if rst.proc_num > 1 and not hasattr(ctx, "run_steps_dist"):
            limit = False   # (the Python orchestration of the rehearsals, distributed.PhasedStepper, does not observe the device-side limit)
        try:
            steps0, first = int(vs.itt), True
            while True:
                now = int(vs.time)
                if now >= t_stop:
                    break
                # a round ends where the run ends, or where the output intervals it may start would not fit the device's slots
                t_round = t_stop if slots is None else min(t_stop, (now // iv + slots - 1) * iv)
                n = -(-(t_round - now) // 86400)          # a step covers at most a day: this many steps never overshoot
                if limit:
                    vs.flush_to_device()
                    ctx.set_time_limit(t_round)
                    done = int(vs.itt) - steps0
                    if not first and done > 0:             # what the mean step length so far suggests, and a margin
                        n = max(n, int((t_round - now) / max(600.0, (now - start_time) / done) * 1.1) + 8)
                elif slots is not None:
                    n = min(n, max(1, slots - 1))          # (a step starts at most one output interval)
                self.run_device(int(n), final=False)
                first = False
        finally:
            if limit:
                ctx.set_time_limit(None)

    def _lean_host_loop_possible(self):
        settings = self.state.settings
        return (self._fused_host_step_possible() and not rs.profile_mode and settings.restart_frequency <= 0
                and hasattr(self.state.backend_context, "step_scalars") and not os.environ.get("RH_NO_LEAN_LOOP"))

    def _run_host_hooks(self, start_time, runlen):
        """`while vs.time - start_time < runlen: step()` (roger/roger.py:548-556) for a script whose hooks in FRONT of the physics are
        its own: those run on the host, step by step, as in the reference; the rest of the step is ONE native call that also brings back
        the scalars the loop and the hooks look at (rh_svat_step_scalars: no second call, no staged copies).  What step() does around
        the hooks per step -- five timer contexts, the routine wrappers' unlock, a list of 200 names to mark, three device read-backs
        -- was 75 - 85 us per step, three times the fused kernel's time on a catchment-sized grid; what is left is the hooks' own work.
        Hooks the probes found to do nothing (hooks.py) are not called."""
        state = self.state
        vs, ctx = state.variables, state.backend_context
        classes = self.hook_classes()
        front = [getattr(getattr(type(self), h), "__wrapped__", getattr(type(self), h))
                 for h in ("read_data", "set_boundary_conditions", "set_forcing")
                 if not (classes[h] and h != "set_forcing")]
        diag = bool(getattr(state, "_diag_active", None))
        timer = state.timers["main"]
        with vs.unlock(), timer:
            s = vs._get_scalars()
            while s.time - start_time < runlen:
                for hook in front:
                    hook(self, state)
                if vs._scalars_dirty or vs._scalars is None:
                    s = vs._get_scalars()       # (the hook assigned a scalar: its own copy is the current one)
                monthly = (s.month[1] != s.month[0]) and s.itt > 1   # the stock set_parameters' test (models/svat.py)
                vs.flush_to_device()
                s = ctx.step_scalars(monthly)
                vs.mark_device_newer(None, scalars=s)
                if diag:
                    self._end_of_step(state)

    def run(self, show_progress_bar=None):
        """roger/roger.py:523-580"""
        self._ensure_setup_done()
        vs = self.state.variables
        settings = self.state.settings
        runlen = settings.runlen if settings.warmup_done else settings.runlen_warmup   # roger/roger.py:541-546
        start_time = vs.time
        try:
            if self.device_run_possible():
                self._run_on_device(int(start_time), int(runlen))
            elif self._lean_host_loop_possible():
                self._run_host_hooks(int(start_time), int(runlen))
            else:
                while vs.time - start_time < runlen:
                    self.step(self.state)
        except BaseException:
            # the forced restart below is a collective on several ranks (the slabs are gathered to rank 0): a rank that leaves run() on
            # an exception would wait there for peers that are still stepping -- the job would hang instead of failing
            failed = True
            raise
        else:
            failed = False
        finally:
            if settings.write_restart and not settings.enable_offline_transport and not (failed and rst.proc_num > 1):   # roger/roger.py:577-579
                restart.write_restart(self.state, force=True)
        (self.state.sas_context or self.state.backend_context).sync()
        diagnostics.close(self.state)

    # -- fast path --------------------------------------------------------------------------------
    def enable_device_hooks(self):
        """Hand the forcing series of set_forcing_setup (vs.PREC, vs.TA, vs.PET, vs.YEAR, vs.MONTH,
        vs.DOY) to the device; `run_device` then performs the benchmark's set_forcing /
        set_parameters hooks there."""
        self._ensure_setup_done()
        import numpy as np

        vs = self.state.variables
        ctx = self.state.backend_context
        vs.flush_to_device()
        weights = [np.asarray(getattr(vs, k))[2:-2, 2:-2].reshape(-1) for k in ("prec_weight", "ta_offset", "pet_weight")]
        neutral = (weights[0] == 1).all() and (weights[1] == 0).all() and (weights[2] == 1).all()
        if self.state.settings.enable_distributed_input:
            # several stations: vs.PREC_DIST / TA_DIST / PET_DIST (n_stations, t_forc), vs.station_id per cell, vs.station_ids
            # (roger/bmimodels/svat_dist/svat_dist.py:200-211, 261-263, 280-293)
            ids = np.asarray(vs.station_ids)
            cell = np.asarray(vs.station_id)[2:-2, 2:-2].reshape(-1)
            index = np.array([int(np.where(ids == c)[0][0]) if c in ids else -1 for c in cell], dtype=np.int32)
            ctx.set_forcing_stations(dict(PREC=vs.PREC_DIST, TA=vs.TA_DIST, PET=vs.PET_DIST, YEAR=vs.YEAR, MONTH=vs.MONTH, DOY=vs.DOY), index)
            ctx.set_forcing_weights(*weights)
        else:
            ctx.set_forcing_series(dict(PREC=vs.PREC, TA=vs.TA, PET=vs.PET, YEAR=vs.YEAR, MONTH=vs.MONTH, DOY=vs.DOY))
            if not neutral:   # eberbaechle/svat_distributed/svat.py:169-186, 276-296
                ctx.set_forcing_weights(*weights)
        self._device_hooks = True
        self._per_cell_forcing = bool(self.state.settings.enable_distributed_input or not neutral)

    def _stepper(self, one_exchange):
        key = "_stepper_one" if one_exchange else "_stepper_three"
        if getattr(self, key, None) is None:
            setattr(self, key, distributed.PhasedStepper(distributed.phases_for(self.state.backend_context, one_exchange=one_exchange)))
        return getattr(self, key)

    def run_device(self, nsteps, final=True):
        if not getattr(self, "_device_hooks", False):
            self.enable_device_hooks()
        vs = self.state.variables
        vs.flush_to_device()
        ctx = self.state.backend_context
        if rst.proc_num > 1 and getattr(self, "_per_cell_forcing", False):
            # per-cell forcing (station weights / several stations): every column forms its own prec / ta, so both predicate words are
            # evaluated over the columns and exchanged -- the three-phase protocol (rh_run_steps_dist and the summary path need
            # forcing shared by all columns and say so with RH_ERR_STATE)
            self._stepper(one_exchange=False).run(nsteps)
        elif rst.proc_num > 1:
            # several ranks: one exchange of the summary word per step -- from C over RCCL where the context offers it
            # (rh_comm_init + rh_run_steps_dist), through torch.distributed otherwise
            if hasattr(ctx, "run_steps_dist"):
                if not getattr(self, "_comm_ready", False):
                    ctx.comm_init_torch()
                    self._comm_ready = True
                ctx.run_steps_dist(nsteps)
            else:
                self._stepper(one_exchange=True).run(nsteps)
        else:
            ctx.run_steps(nsteps)
        vs.mark_device_newer()
        if getattr(self.state, "_diag_active", None):
            diagnostics.output(self.state, final=final)
