"""Run log of a training job: one JSON object per line in `{basedir}/{expname}/log.jsonl`.

The reference appends `loss.item()` to a Python list every iteration and plots it next to the rendered frame at `i_render`
(`mlx_nerf/entrypoints/__test_nerf.py:298-299,314-322`); headless, the same information goes to a file a plot can be made
from later.  Records (rank 0 writes; every record carries `"kind"`):

    {"kind": "run",   "time", "world_size", "n_rand", "precision", "resumed_from", ...}             once, at start / resume
    {"kind": "train", "it", "loss_coarse", "loss_fine", "psnr_coarse", "psnr_fine", "lr", "rays_per_s", "elapsed_s"}
                                                                                                   every `log_every` iterations
    {"kind": "eval",  "it", "psnr", "view", "seconds"}                                             every `render_every` iterations

`rays_per_s` is the WHOLE job's training rate (rays of all ranks) over the iterations since the previous train record, wall
clock, taken after the loss values have reached the host (which waits for the device).  Pure host code: no GPU needed.
"""
import json
import math
import os
import time
from typing import Optional


def _num(x) -> Optional[float]:
    """JSON has no NaN / Infinity: non-finite values are written as null (and counted by the reader as such)."""
    if x is None:
        return None
    x = float(x)
    return x if math.isfinite(x) else None


def psnr_of_mse(mse: float) -> Optional[float]:
    """`PSNR = -10 log10(MSE)` (ops/metric.py:11-16 of the reference), None for a non-positive or non-finite MSE."""
    mse = float(mse)
    if not (mse > 0.0) or not math.isfinite(mse):
        return None
    return -10.0 * math.log10(mse)


class RunLog:
    def __init__(self, directory: str, rank: int = 0, world: int = 1, name: str = "log.jsonl", enabled: bool = True):
        self.path = os.path.join(directory, name)
        self.world = int(world)
        self.active = bool(enabled) and int(rank) == 0
        self._t0 = time.perf_counter()
        self._last_t, self._last_it, self._pause = self._t0, None, 0.0
        if self.active:
            os.makedirs(directory, exist_ok=True)

    def _write(self, rec: dict):
        if not self.active:
            return
        with open(self.path, "a") as fp:                       # append: a resumed run continues the same file
            fp.write(json.dumps(rec) + "\n")

    def run(self, **fields):
        self._write({"kind": "run", "time": time.strftime("%Y-%m-%dT%H:%M:%S"), "world_size": self.world, **fields})

    def mark(self, it: int):
        """Start of the rate window (call once before the loop with the iteration the loop resumes from)."""
        self._last_t, self._last_it, self._pause = time.perf_counter(), int(it), 0.0

    def train(self, it: int, loss_coarse, loss_fine, lr: float, n_rand: int) -> dict:
        now = time.perf_counter()
        its = None if self._last_it is None else int(it) - self._last_it
        dt = now - self._last_t - self._pause                            # evaluation renders inside the window are not training time
        rate = None if not its or dt <= 0 else its * int(n_rand) * self.world / dt
        rec = {"kind": "train", "it": int(it), "loss_coarse": _num(loss_coarse), "loss_fine": _num(loss_fine),
               "psnr_coarse": None if loss_coarse is None else psnr_of_mse(loss_coarse),
               "psnr_fine": None if loss_fine is None else psnr_of_mse(loss_fine),
               "lr": float(lr), "rays_per_s": _num(rate), "elapsed_s": now - self._t0}
        self._write(rec)
        self._last_t, self._last_it, self._pause = time.perf_counter(), int(it), 0.0      # the write is not part of the next window
        return rec

    def eval(self, it: int, psnr, view, seconds: float) -> dict:
        rec = {"kind": "eval", "it": int(it), "psnr": _num(psnr), "view": view, "seconds": float(seconds)}
        self._write(rec)
        self._pause += float(seconds)                                   # a rendered frame is not training time
        return rec


def read(path: str):
    """All records of a log file (blank lines skipped)."""
    with open(path) as fp:
        return [json.loads(ln) for ln in fp if ln.strip()]
