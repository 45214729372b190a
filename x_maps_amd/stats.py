"""Minimal stand-in for the reference's StatsPrinter (python/stats_printer.py:162-306): the same call
surface the hot path uses -- count / add_metric / measure_time / log -- without the ANSI terminal UI
(out of scope).  Stage names are the reference's ("ev rect", "x-maps disp", "disp map", ...)."""
from __future__ import annotations

import time
from collections import defaultdict
from contextlib import contextmanager


class StatsPrinter:
    def __init__(self):
        self.counters = defaultdict(int)
        self.metrics = defaultdict(list)
        self.timers = defaultdict(list)
        self.logs = []

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
        for k, v in self.counters.items():
            print(f"{k}: {v}")
        for k, v in self.timers.items():
            print(f"{k}: {1e3 * sum(v) / max(len(v), 1):.3f} ms avg over {len(v)}")

    def toggle_silence(self):
        pass
