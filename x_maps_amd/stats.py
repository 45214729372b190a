"""Minimal stand-in for the reference's StatsPrinter (python/stats_printer.py:162-306): the same call
surface the hot path uses -- count / add_metric / measure_time / log -- without the ANSI terminal UI
(out of scope).  Stage names are the reference's ("ev rect", "x-maps disp", "disp map", ...).
Memory is bounded: the reference's printer averages over a local and a global window; here every metric / timer keeps its
count, sum, extrema and the last WINDOW values (a 60 Hz live loop adds several values per frame for as long as it runs)."""
from __future__ import annotations

import time
from collections import defaultdict, deque
from contextlib import contextmanager

WINDOW = 256  # values kept per key (the "local" window); older ones live on in count / total / min / max


class Series:
    """running record of one metric or timer"""
    __slots__ = ("count", "total", "min", "max", "recent")

    def __init__(self):
        self.count, self.total, self.min, self.max = 0, 0.0, float("inf"), float("-inf")
        self.recent = deque(maxlen=WINDOW)

    def append(self, v):
        self.count += 1
        self.total += v
        self.min = v if v < self.min else self.min
        self.max = v if v > self.max else self.max
        self.recent.append(v)

    def mean(self):
        return self.total / self.count if self.count else 0.0

    def __len__(self):
        return self.count

    def __getitem__(self, i):  # (the most recent values, list-like: series[-1])
        return self.recent[i]


class StatsPrinter:
    def __init__(self):
        self.counters = defaultdict(int)
        self.metrics = defaultdict(Series)
        self.timers = defaultdict(Series)
        self.logs = deque(maxlen=WINDOW)
        self.silent = False

    def count(self, key, n=1):
        self.counters[key] += n

    def add_metric(self, key, value):
        self.metrics[key].append(value)

    def log(self, msg):
        self.logs.append(msg)

    @contextmanager
    def measure_time(self, key):
        t0 = time.perf_counter()
        try:
            yield
        finally:
            self.timers[key].append(time.perf_counter() - t0)

    def print_stats_if_needed(self):
        pass

    def print_stats(self):
        if self.silent:
            return
        for k, v in self.counters.items():
            print(f"{k}: {v}")
        for k, v in self.timers.items():
            print(f"{k}: {1e3 * v.mean():.3f} ms avg over {v.count}")

    def toggle_silence(self):
        self.silent = not self.silent
