"""Env-gated step profiler and ETA timer (reference vision_base/utils/timer.py:5-89)."""
import os
import time
from functools import wraps

import torch


def profile(name, profile_start=0, profile_end=1):
    def deco(func):
        debugging = os.environ.get("DEBUGGING", "0").lower() in ("1", "true")
        if not debugging:
            return func
        state = {"n": 0}

        @wraps(func)
        def wrapped(*a, **k):
            state["n"] += 1
            if profile_start <= state["n"] - 1 < profile_end:
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
                t0 = time.time()
                out = func(*a, **k)
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
                print("%s call %d: %.3f ms" % (name, state["n"], 1e3 * (time.time() - t0)))
                return out
            return func(*a, **k)
        return wrapped
    return deco


class Timer(object):
    def __init__(self):
        self.start = time.time()

    def time_diff_per_n_loops(self):
        return time.time() - self.start

    def compute_eta(self, current_iter, total_iter):
        elapsed = time.time() - self.start
        if current_iter <= 0:
            return "n/a"
        left = elapsed / current_iter * max(total_iter - current_iter, 0)
        h, rem = divmod(int(left), 3600)
        m, s = divmod(rem, 60)
        return "%dh:%02dm:%02ds" % (h, m, s)
