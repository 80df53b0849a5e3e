"""Domain decomposition and the per-step predicate exchange for multi-GPU runs.

Decomposition API follows the reference (roger/distributed.py:121-187): `num_proc = (px, py)`,
ranks laid out x-fastest, even divisibility required, chunk slices with or without the 2-cell
overlap, process neighbours, `exchange_overlap` for the 2-cell ghost frame, global reductions.  On
this path columns never read their neighbours, so no halo is exchanged; what the reference does per
step -- gather 18 fields to rank 0, decide dt there, scatter back
(adaptive_time_stepping_dist_safe.py:6-26) -- is replaced by one all-reduce of 64 predicate bits
(as 64 int32, MAX) per step.
"""


def validate_decomposition(nx, ny, num_proc, comm_size):
    """roger/distributed.py:121-138"""
    px, py = num_proc
    if px * py != comm_size:
        raise RuntimeError(f"number of processes ({px * py}) does not match size of communicator ({comm_size})")
    if nx % px:
        raise ValueError("processes do not divide domain evenly in x-direction")
    if ny % py:
        raise ValueError("processes do not divide domain evenly in y-direction")


def get_chunk_size(nx, ny, num_proc):
    return (nx // num_proc[0], ny // num_proc[1])


def proc_rank_to_index(rank, num_proc):
    return (rank % num_proc[0], rank // num_proc[0])


def proc_index_to_rank(ix, iy, num_proc):
    return ix + iy * num_proc[0]


def get_chunk_slices(nx, ny, num_proc, rank, include_overlap=False):
    """(global_slice, local_slice) over the (x, y) dims of a ghosted array; roger/distributed.py:153-187."""
    px, py = proc_rank_to_index(rank, num_proc)
    nxl, nyl = get_chunk_size(nx, ny, num_proc)
    if include_overlap:
        sxl = 0 if px == 0 else 2
        sxu = nxl + 4 if (px + 1) == num_proc[0] else nxl + 2
        syl = 0 if py == 0 else 2
        syu = nyl + 4 if (py + 1) == num_proc[1] else nyl + 2
    else:
        sxl = syl = 0
        sxu, syu = nxl, nyl
    return ((slice(sxl + px * nxl, sxu + px * nxl), slice(syl + py * nyl, syu + py * nyl)),
            (slice(sxl, sxu), slice(syl, syu)))


def get_process_neighbors(rank, num_proc):
    """Ranks of the eight neighbours (None at the domain edge); roger/distributed.py:190-220."""
    px, py = proc_rank_to_index(rank, num_proc)
    west = px - 1 if px != 0 else None
    east = px + 1 if px != num_proc[0] - 1 else None
    south = py - 1 if py != 0 else None
    north = py + 1 if py != num_proc[1] - 1 else None
    idx = dict(west=(west, py), south=(px, south), east=(east, py), north=(px, north), southwest=(west, south),
               southeast=(east, south), northeast=(east, north), northwest=(west, north))
    return {k: (proc_index_to_rank(*i, num_proc) if None not in i else None) for k, i in idx.items()}


_SEND_ORDER = ("west", "northwest", "north", "northeast", "east", "southeast", "south", "southwest")
_RECV_ORDER = ("east", "southeast", "south", "southwest", "west", "northwest", "north", "northeast")
_FROM = dict(west=(slice(2, 4), slice(None)), south=(slice(None), slice(2, 4)), east=(slice(-4, -2), slice(None)),
             north=(slice(None), slice(-4, -2)), southwest=(slice(2, 4), slice(2, 4)), southeast=(slice(-4, -2), slice(2, 4)),
             northeast=(slice(-4, -2), slice(-4, -2)), northwest=(slice(2, 4), slice(-4, -2)))
_TO = dict(west=(slice(0, 2), slice(None)), south=(slice(None), slice(0, 2)), east=(slice(-2, None), slice(None)),
           north=(slice(None), slice(-2, None)), southwest=(slice(0, 2), slice(0, 2)), southeast=(slice(-2, None), slice(0, 2)),
           northeast=(slice(-2, None), slice(-2, None)), northwest=(slice(0, 2), slice(-2, None)))


def exchange_overlap(arr, num_proc, group=None):
    """Fill the 2-cell ghost frame of an (x, y, ...) array from the neighbouring ranks' interiors, non-periodic
    (roger/distributed.py:223-331, the (x, y) case; same send / receive order).  `arr` is a torch tensor on the
    rank's device -- RCCL moves it over xGMI between GPUs, gloo between CPU ranks -- or a numpy array (a new array
    is returned).  Nothing on the SVAT / oneD / transport paths of the BASELINE configs calls this (columns do not
    read their neighbours, routing is off); it completes the decomposition API for setups that do."""
    import numpy as np
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return arr
    is_np = isinstance(arr, np.ndarray)
    t = torch.from_numpy(np.array(arr)) if is_np else arr
    nb = get_process_neighbors(dist.get_rank(group), num_proc)
    for send_dir, recv_dir in zip(_SEND_ORDER, _RECV_ORDER):
        send_proc, recv_proc = nb[send_dir], nb[recv_dir]
        if send_proc is None and recv_proc is None:
            continue
        ops, recv_buf = [], None
        if send_proc is not None:
            ops.append(dist.P2POp(dist.isend, t[_FROM[send_dir]].contiguous(), send_proc, group))
        if recv_proc is not None:
            recv_buf = torch.empty_like(t[_TO[recv_dir]]).contiguous()
            ops.append(dist.P2POp(dist.irecv, recv_buf, recv_proc, group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        if recv_buf is not None:
            t[_TO[recv_dir]] = recv_buf
    return t.numpy() if is_np else t


def _reduce(value, op, group=None):
    """Scalar reduction over the ranks; roger/distributed.py:357-416."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return value
    t = torch.tensor([float(value)], dtype=torch.float64)
    if dist.get_backend(group) == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=op, group=group)
    return type(value)(t.item()) if isinstance(value, (bool, int, float)) else t.item()


def global_and(value, group=None):
    import torch.distributed as dist
    return bool(_reduce(float(bool(value)), dist.ReduceOp.MIN, group))


def global_or(value, group=None):
    import torch.distributed as dist
    return bool(_reduce(float(bool(value)), dist.ReduceOp.MAX, group))


def global_max(value, group=None):
    import torch.distributed as dist
    return _reduce(value, dist.ReduceOp.MAX, group)


def global_min(value, group=None):
    import torch.distributed as dist
    return _reduce(value, dist.ReduceOp.MIN, group)


def global_sum(value, group=None):
    import torch.distributed as dist
    return _reduce(value, dist.ReduceOp.SUM, group)


class PhasedStepper:
    """Runs whole time steps of a backend that exposes the three-phase step, all-reducing the two
    predicate words over `group` between the phases.

    `backend` needs: hooks_phase(), phase1(), phase2(), phase3(), predicate_buffer(word) -> a
    torch tensor of 64 int32 holding 0/1, and load_predicates(word, tensor).  The HIP backend
    (HipPhases below) and the CPU test double used by the gloo tests both provide them.
    """

    def __init__(self, backend, group=None, always_exchange=False):
        self.backend = backend
        self.group = group
        self.always_exchange = always_exchange   # tests: go through the exchange buffers also with one rank

    def _exchange(self, word):
        import torch.distributed as dist

        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1
        if not (multi or self.always_exchange):
            return  # single rank: the local predicate words already are the global ones
        buf = self.backend.predicate_buffer(word)
        if multi:
            dist.all_reduce(buf, op=dist.ReduceOp.MAX, group=self.group)
        self.backend.load_predicates(word, buf)

    def step(self):
        b = self.backend
        if getattr(b, "one_exchange", False):
            # summary path (shared forcing): both predicate words follow from one summary word, include/roger_hip.h
            if getattr(b, "fused_exchange", False):
                # the exchange format written / read by the summary and control kernels themselves: per step one
                # reduction kernel, one 256-byte all-reduce, the control kernel and the fused kernel
                import torch.distributed as dist

                buf = b.summary_to_buffer()
                if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
                    dist.all_reduce(buf, op=dist.ReduceOp.MAX, group=self.group)
                b.finish_from_buffer(buf)
                return
            b.summary_phase()
            self._exchange(3)
            b.finish_phase()
            return
        b.hooks_phase()
        b.phase1()
        self._exchange(0)
        b.phase2()
        self._exchange(1)
        b.phase3()

    def run(self, nsteps):
        for _ in range(nsteps):
            self.step()

    def adaptive_dt(self):
        """The stand-alone adaptive time stepping (rh_adaptive_dt) of the hook-preserving `RogerSetup.step()` with several ranks: the
        two predicate words are all-reduced between its parts, so that every rank derives the same dt, event ids and forcing window
        (the reference gathers 18 fields to rank 0 for it, adaptive_time_stepping_dist_safe.py:6-26)."""
        b = self.backend
        b.phase1()
        self._exchange(0)
        b.phase2()
        self._exchange(1)
        b.adaptive_dt_finish()


def phases_for(ctx, one_exchange=True):
    """The PhasedStepper backend of a context: HipPhases on the context's device for `_native.Context`; a context that brings its
    own (the tests' device double) says so through `make_phases`."""
    if hasattr(ctx, "make_phases"):
        return ctx.make_phases(one_exchange=one_exchange)
    import os

    import torch

    return HipPhases(ctx, torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))), one_exchange=one_exchange)


class HipPhases:
    """Adapter of a `_native.Context` to PhasedStepper; the exchange buffers are torch tensors on
    the context's device and all launches go to torch's current stream so that the RCCL
    all-reduce is stream-ordered with the kernels."""

    def __init__(self, ctx, device, one_exchange=True, fused_exchange=True):
        import torch

        self.ctx = ctx
        self.one_exchange = one_exchange   # False: the three-phase protocol (needed with per-cell forcing)
        # the summary / control kernels write and read the 64-int32 exchange format themselves (False: separate
        # rh_predicates_expand / rh_predicates_compress launches, the generic route also used by the three-phase protocol)
        self.fused_exchange = fused_exchange
        self.buf = {w: torch.zeros(64, dtype=torch.int32, device=device) for w in (0, 1, 3)}
        ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)

    def summary_phase(self):
        self.ctx.call("rh_step_summary")   # device-side hooks included

    def finish_phase(self):
        self.ctx.step_finish(-1)

    def summary_to_buffer(self):
        self.ctx.step_summary_expand(self.buf[3].data_ptr())
        return self.buf[3]

    def finish_from_buffer(self, tensor):
        self.ctx.step_finish_compress(tensor.data_ptr(), -1)

    def hooks_phase(self):
        self.ctx.call("rh_hooks_phase")

    def phase1(self):
        self.ctx.call("rh_step_phase1")

    def phase2(self):
        self.ctx.call("rh_step_phase2")

    def phase3(self):
        self.ctx.step_phase3(-1)

    def adaptive_dt_finish(self):
        self.ctx.call("rh_adaptive_dt_finish")

    def predicate_buffer(self, word):
        self.ctx.predicates_expand(word, self.buf[word].data_ptr())
        return self.buf[word]

    def load_predicates(self, word, tensor):
        self.ctx.predicates_compress(word, tensor.data_ptr())
