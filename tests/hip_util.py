"""Test-side helpers around roger_amd._native.Context (GPU tests)."""
import numpy as np

from roger_amd import _native as N

SCALARS = ("itt", "time", "dt_secs", "itt_day", "itt_forc", "time_event0", "event_id_counter", "dt")
SCALARS2 = ("event_id", "year", "month", "doy")


def _held(ctx, nm):
    """The routing's planes exist in routing contexts only (rh_planes_held); the oracle double holds every plane."""
    return nm in ctx.index and ctx.index[nm] < getattr(ctx, "planes_held", len(ctx.index))


def upload_snapshot(ctx, snap, names):
    for row, nm in zip(snap, names):
        if _held(ctx, nm):
            ctx.upload(nm, row)


def download_snapshot(ctx, names):
    return np.stack([ctx.download(nm).astype(np.float64) if _held(ctx, nm) else np.zeros(ctx.n) for nm in names])


def scalars_from_row(row):
    s = N.RhScalars()
    i = 0
    for k in SCALARS:
        setattr(s, k, float(row[i]) if k == "dt" else int(row[i]))
        i += 1
    for k in SCALARS2:
        getattr(s, k)[0] = int(row[i])
        getattr(s, k)[1] = int(row[i + 1])
        i += 2
    s.sanity_ok = 1
    return s


def scalars_to_row(s):
    row = [float(getattr(s, k)) for k in SCALARS]
    for k in SCALARS2:
        row += [float(getattr(s, k)[0]), float(getattr(s, k)[1])]
    return np.array(row)


class HipForcingDriver:
    """The `set_forcing` / `set_parameters` user hooks of the benchmark setup
    (benchmarks/SVAT_benchmark.py:105-110,151-171) driving the C ABI directly."""

    def __init__(self, ctx, forcing, per_cell=False):
        self.ctx, self.F, self.per_cell = ctx, forcing, per_cell
        # host mirror of the few scalars the hooks need; kept in step with the device by
        # replaying the same integer bookkeeping from rh_get_scalars when asked
        self.s = None

    def before_step(self):
        s = self.ctx.get_scalars()
        if s.time % 86400 == 0:
            i = s.itt_forc
            s.itt_day = 0
            s.year[1] = int(self.F["YEAR"][i])
            s.month[1] = int(self.F["MONTH"][i])
            s.doy[1] = int(self.F["DOY"][i])
            s.itt_forc = i + 144
            self.ctx.set_scalars(s)
            day = [self.F[k][i:i + 144] for k in ("PREC", "TA", "PET")]
            if self.per_cell:
                day = [np.broadcast_to(d, (self.ctx.n, 144)).copy() for d in day]
            self.ctx.set_forcing_day(*day)
        return (s.month[1] != s.month[0]) and (s.itt > 1)
