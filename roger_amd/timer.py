"""roger/timer.py: wall-clock timer used for state.timers[...]."""
import timeit


class Timer:
    def __init__(self):
        self.total_time = 0
        self.last_time = 0

    def __enter__(self):
        self.start_time = timeit.default_timer()

    def __exit__(self, *args):
        self.last_time = timeit.default_timer() - self.start_time
        self.total_time += self.last_time
