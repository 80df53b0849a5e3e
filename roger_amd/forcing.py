"""Synthetic 10-minute forcing for the SVAT path.

Follows the *recipe* of the reference's toy-data generator
(roger/tools/make_toy_data.py:17-170: seeded ``default_rng(42)``, uniform rain
bursts at slot 12 and at half the series, daily TA/PET forward-filled to 10 min,
``PET/24/6``) but returns in-memory arrays instead of writing ``forcing.nc``
(the benchmark inputs ``benchmarks/input/**.nc`` are not shipped by the
reference).  Used by the golden-vector generator, the parity tests and bench.py,
so all three see the same numbers.
"""
import numpy as np

SLOTS_PER_DAY = 6 * 24


def _daily_to_10min(x):
    return np.repeat(np.asarray(x, dtype=np.float64), SLOTS_PER_DAY)


def toy_forcing(event_type="rain", ndays=10, seed=42):
    """Return dict(PREC, TA, PET, YEAR, MONTH, DOY), each of length ndays*144.

    PREC in mm/10 min, TA in degC, PET in mm/10 min.  Calendar starts 2018-01-01
    as in the reference generator.
    """
    rng = np.random.default_rng(seed)
    n = ndays * SLOTS_PER_DAY
    prec = np.zeros(n)
    if event_type in ("rain", "snow", "snow+rain"):
        burst = rng.uniform(0, 1, 18)
    elif event_type == "heavyrain":
        burst = rng.uniform(0.1, 6, 12 * 6)
    elif event_type == "norain":
        burst = np.zeros(0)
    else:
        raise ValueError(f"unknown event_type {event_type!r}")
    if burst.size:
        prec[12 : 12 + burst.size] = burst
        h = int(n / 2)
        prec[h : h + burst.size] = burst[: max(0, min(burst.size, n - h))]
    if event_type in ("rain", "heavyrain", "norain"):
        ta = rng.uniform(15, 20, ndays)
        pet = rng.uniform(2, 3, ndays)
    elif event_type == "snow":
        ta = rng.uniform(-3, -1, ndays)
        pet = rng.uniform(1, 2, ndays)
    else:  # snow+rain
        ta = rng.uniform(0, 3, ndays)
        ta[:2] = -1
        pet = rng.uniform(1, 2, ndays)
    return _pack(prec, ta, pet, ndays)


def _pack(prec, ta_daily, pet_daily, ndays, start="2018-01-01"):
    days = np.datetime64(start) + np.arange(ndays)
    years = days.astype("datetime64[Y]").astype(int) + 1970
    months = days.astype("datetime64[M]").astype(int) % 12 + 1
    doy = (days - days.astype("datetime64[Y]")).astype(int) + 1
    return dict(
        PREC=np.asarray(prec, dtype=np.float64),
        TA=_daily_to_10min(ta_daily),
        PET=_daily_to_10min(pet_daily) / 24 / 6,
        YEAR=np.repeat(years, SLOTS_PER_DAY).astype(np.int64),
        MONTH=np.repeat(months, SLOTS_PER_DAY).astype(np.int64),
        DOY=np.repeat(doy, SLOTS_PER_DAY).astype(np.int64),
    )


def combo_forcing(ndays=30, seed=42, start="2018-01-20"):
    """A seeded series that exercises every dt class of the adaptive stepper.

    Cold days with snowfall, a melt period, moderate rain (hourly steps), heavy
    rain above hpi = 5 mm/10 min (10-minute steps), rain pauses inside an event,
    and dry days (daily steps).  Crosses a month boundary so the monthly surface
    parameter update runs.
    """
    rng = np.random.default_rng(seed)
    n = ndays * SLOTS_PER_DAY
    prec = np.zeros(n)
    ta = np.empty(ndays)
    pet = np.empty(ndays)
    for d in range(ndays):
        phase = d % 10
        if phase in (0, 1):  # snowfall
            ta[d] = rng.uniform(-4, -0.5)
            pet[d] = rng.uniform(0.3, 1.0)
            k = d * SLOTS_PER_DAY + int(rng.integers(6, 60))
            prec[k : k + 30] = rng.uniform(0, 0.6, 30)
        elif phase == 2:  # melt, no precipitation
            ta[d] = rng.uniform(1, 4)
            pet[d] = rng.uniform(0.5, 1.5)
        elif phase in (3, 6):  # moderate rain with a pause
            ta[d] = rng.uniform(6, 14)
            pet[d] = rng.uniform(1.5, 3)
            k = d * SLOTS_PER_DAY + int(rng.integers(0, 40))
            prec[k : k + 12] = rng.uniform(0, 1.2, 12)
            prec[k + 30 : k + 48] = rng.uniform(0, 0.9, 18)
        elif phase == 4:  # heavy rain
            ta[d] = rng.uniform(12, 20)
            pet[d] = rng.uniform(2, 4)
            k = d * SLOTS_PER_DAY + int(rng.integers(20, 70))
            prec[k : k + 24] = rng.uniform(0.1, 9, 24)
        else:  # dry
            ta[d] = rng.uniform(8, 22)
            pet[d] = rng.uniform(2, 4.5)
    return _pack(prec, ta, pet, ndays, start=start)


# ---------------------------------------------------------------------------------------------------------------------
# forcing from the reference's text inputs (host part)
# ---------------------------------------------------------------------------------------------------------------------
def _read_table(path, column):
    """One of PREC.txt / TA.txt / PET.txt: whitespace-separated, header `YYYY MM DD hh mm <column> ...`, -9999 = missing
    (roger/io_tools/csv.py:10-104)."""
    import pandas as pd

    df = pd.read_csv(path, sep=r"\s+", header=0, na_values=-9999)
    index = pd.to_datetime(dict(year=df.YYYY, month=df.MM, day=df.DD, hour=df.hh, minute=df.mm))
    return pd.Series(df[column].to_numpy(dtype=np.float64), index=index, name=column)


def forcing_from_txt(input_dir, float_type="float32", ndays=None):
    """10-minute forcing arrays from a directory with PREC.txt (10-minute sums), TA.txt and PET.txt (daily values) --
    what `read_meteo` + `write_forcing` put into forcing.nc and the setup scripts read back
    (roger/io_tools/csv.py:10-104, roger/tools/setup.py:469-620):

      * precipitation on a gap-free 10-minute axis from 00:00 of its first day to 23:50 of its last, missing slots 0;
      * air temperature interpolated linearly over missing values, then it and PET joined to the precipitation axis and
        forward-filled; PET / 24 / 6 (mm per 10 minutes);
      * values rounded through `float_type` (write_forcing stores float32 unless told otherwise) and returned as float64;
      * YEAR / MONTH / DOY of every slot.

    Returns the dict that `Context.set_forcing_series` and the setup hooks take (PREC, TA, PET, YEAR, MONTH, DOY);
    `ndays` keeps the first days only."""
    import os

    import pandas as pd

    prec = _read_table(os.path.join(input_dir, "PREC.txt"), "PREC")
    ta = _read_table(os.path.join(input_dir, "TA.txt"), "TA").interpolate(method="linear", limit_direction="forward")
    pet = _read_table(os.path.join(input_dir, "PET.txt"), "PET")
    first, last = prec.index[0].normalize(), prec.index[-1].normalize()
    axis = pd.date_range(first, last + pd.Timedelta(hours=23, minutes=50), freq="10min")
    table = pd.DataFrame({"PREC": 0.0}, index=axis)
    table.loc[prec.index, "PREC"] = prec.to_numpy()
    table = table.join([ta.to_frame(), pet.to_frame()]).ffill()
    for col in ("PREC", "TA", "PET"):
        bad = ~np.isfinite(table[col].to_numpy())
        if bad.any():
            raise ValueError(f"{col}: {int(bad.sum())} non-numeric values (first at {table.index[bad][0]})")
    table["PET"] = (table["PET"] / 24) / 6
    if ndays is not None:
        table = table.iloc[: int(ndays) * SLOTS_PER_DAY]
    ft = np.dtype(float_type)
    out = {k: table[k].to_numpy().astype(ft).astype(np.float64) for k in ("PREC", "TA", "PET")}
    out["YEAR"] = table.index.year.to_numpy().astype(np.int64)
    out["MONTH"] = table.index.month.to_numpy().astype(np.int64)
    out["DOY"] = table.index.dayofyear.to_numpy().astype(np.int64)
    return out


def forcing_from_nc(path, ndays=None, cell=(0, 0)):
    """The same dict from a `forcing.nc` as the reference's `write_forcing` stores it (roger/tools/setup.py:565-626: netCDF-4 with
    PREC / TA / PET (x, y, Time) -- one station broadcast over the grid --, YEAR / MONTH / DOY (Time)) and its setup scripts read it
    back (`_read_var_from_nc(...)[0, 0, :]`, e.g. examples/hillslope_scale/oneD_distributed_routing_tutorial/oneD.py:500-504).  Read with
    h5py where it is installed, otherwise with `roger_amd.h5lite`."""
    try:
        import h5py

        with h5py.File(path, "r") as f:
            v = {k: np.asarray(f[k]) for k in ("PREC", "TA", "PET", "YEAR", "MONTH", "DOY")}
    except ImportError:
        from . import h5lite

        v = h5lite.read_root(path)
    missing = [k for k in ("PREC", "TA", "PET", "YEAR", "MONTH", "DOY") if k not in v]
    if missing:
        raise KeyError(f"{path} lacks the variables {missing}")
    n = v["PREC"].shape[-1] if ndays is None else min(int(ndays) * SLOTS_PER_DAY, v["PREC"].shape[-1])
    out = {k: np.asarray(v[k][cell[0], cell[1], :n], dtype=np.float64) for k in ("PREC", "TA", "PET")}
    out.update({k: np.asarray(v[k][:n], dtype=np.int64) for k in ("YEAR", "MONTH", "DOY")})
    return out
