"""Wall-clock accumulators behind `state.timers[...]` (the reference's roger/timer.py interface: a context manager with
`last_time` and `total_time` in seconds).  Device work is asynchronous: a timer around a native call measures the time to
issue it unless `runtime_settings.profile_mode` makes the caller synchronise."""
import time


class Timer:
    __slots__ = ("total_time", "last_time", "calls", "_t0")

    def __init__(self):
        self.total_time = 0.0
        self.last_time = 0.0
        self.calls = 0
        self._t0 = None

    def __enter__(self):
        self._t0 = time.perf_counter()
        return self

    def __exit__(self, exc_type, exc, tb):
        elapsed = time.perf_counter() - self._t0
        self.last_time = elapsed
        self.total_time += elapsed
        self.calls += 1
        return False

    def __repr__(self):
        return f"Timer(total={self.total_time:.6f} s, calls={self.calls})"
