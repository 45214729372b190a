"""A wall-clock + SIGTERM guard around the parts of bench.py in which several ranks wait for one another.

The replicas' measurement (BASELINE configs[1], frames across ranks) needs no collective but the timing barrier; the extra leg
behind it (one C-10M frame sharded over the ranks: RCCL communicators, collectives per frame) has only ever run in worlds of one
rank before the driver's multi-GPU run.  A collective that never returns there must not cost the line that is already measured:

    with LegGuard(seconds, on_expire):        # on_expire(reason) runs on the guard's own thread ...
        leg()                                 # ... while the main thread may sit in a C call that never returns

`on_expire` prints what there is to print; the guard then leaves the process with os._exit (no Python shutdown: that would join
the stuck thread / destroy the stuck communicator).  Two triggers: the deadline, and SIGTERM (torch.distributed.run terminates
the surviving ranks when one of them dies) -- the signal's C-level handler writes to a wake-up pipe the guard thread waits on,
so it fires even when the main thread never gets back to the interpreter.
"""
from __future__ import annotations

import os
import select
import signal
import sys
import threading
import time


class LegGuard:
    def __init__(self, seconds: float, on_expire, exit_code: int = 0, name: str = "leg"):
        self.seconds, self.on_expire, self.exit_code, self.name = float(seconds), on_expire, int(exit_code), name
        self._cancel_r, self._cancel_w = os.pipe()
        self._sig_r = self._sig_w = None
        self._old_handler = self._old_fd = None
        self._thread = None
        self.fired = None  # reason, once it has

    def _run(self):
        deadline = time.monotonic() + self.seconds
        fds = [self._cancel_r] + ([self._sig_r] if self._sig_r is not None else [])
        reason = None
        while reason is None:
            left = deadline - time.monotonic()
            if left <= 0:
                reason = f"timeout: {self.name} not finished after {self.seconds:.0f} s"
                break
            ready, _, _ = select.select(fds, [], [], min(left, 1.0))
            if self._cancel_r in ready:
                return
            if self._sig_r is not None and self._sig_r in ready:
                reason = f"terminated: SIGTERM during {self.name} (another rank died?)"
        self.fired = reason
        try:
            self.on_expire(reason)
        except BaseException as e:  # never lose the exit to the report
            sys.stderr.write(f"[bench] guard: on_expire failed: {e!r}\n")
        try:
            sys.stdout.flush()
            sys.stderr.flush()
        finally:
            os._exit(self.exit_code)

    def __enter__(self):
        if threading.current_thread() is threading.main_thread():
            try:
                self._sig_r, self._sig_w = os.pipe()
                os.set_blocking(self._sig_w, False)
                self._old_handler = signal.signal(signal.SIGTERM, lambda *_: None)  # (a Python-level handler must exist for the wake-up fd to be written)
                self._old_fd = signal.set_wakeup_fd(self._sig_w, warn_on_full_buffer=False)
            except (ValueError, OSError):
                self._sig_r = None
        self._thread = threading.Thread(target=self._run, name="bench-guard", daemon=True)
        self._thread.start()
        return self

    def __exit__(self, *exc):
        os.write(self._cancel_w, b"x")
        self._thread.join(timeout=5)
        if self._sig_r is not None:
            try:
                signal.set_wakeup_fd(self._old_fd if self._old_fd is not None else -1)
                signal.signal(signal.SIGTERM, self._old_handler if self._old_handler is not None else signal.SIG_DFL)
            except (ValueError, OSError):
                pass
            os.close(self._sig_r), os.close(self._sig_w)
        os.close(self._cancel_r), os.close(self._cancel_w)
        return False
