#!/usr/bin/env python3
"""Paired PSNR ensemble at equal iterations (north_star: "PSNR within 0.1 dB of the MLX reference at equal iterations").

Training under the reference's formulas is chaotic: one trajectory says nothing about a systematic
difference between two arithmetics.  This tool trains, for MANY seeds, two HIP trainers on IDENTICAL batches (same
images, pixels, importance uniforms, initial weights):

    arm "bf16"  Trainer(precision=16): bf16 MFMA operands, fp32 accumulate -- the benchmarked product path
    arm "fp32"  Trainer(precision=32): the reference's own float32 arithmetic (agrees with the fp32 oracle to 1e-4
                per forward, tests/test_gpu_round2.py), at GPU speed -- the reference-arithmetic arm

and reports, at every checkpoint, the PAIRED statistics of delta_s = PSNR_bf16(seed s) - PSNR_fp32(seed s) on held-out
views: mean, standard deviation, 95 % confidence interval (Student t), sign counts.  For every seed and arm it also
records when the networks are in the "dead-sigma" state (sigma < 0 everywhere -> alpha == 0 -> the loss
equals that of an empty volume and the gradient is exactly zero; Adam's momentum can still carry a network out of it).

`--bridge K` additionally runs, for the first K seeds, the fp32 ORACLE trainer (torch fp32 ops on the device, the same
restatement the CPU tests pin) in lockstep with the fp32 arm, so that "fp32 arm == reference arithmetic" is itself a
measured statement over whole trajectories.

    python tools/psnr_ensemble.py --seeds 32 --hw 100 --n-rand 1024 --iters 2500 --every 250 --bridge 4 --bridge-iters 1500

One JSON line per (seed, checkpoint), per seed summary, and per checkpoint ensemble record.  The oracle is the checker.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_meets_mlx_amd import _native                                # noqa: E402
from nerf_meets_mlx_amd.dataset import synthetic                      # noqa: E402
from nerf_meets_mlx_amd.engine.trainer import Trainer                 # noqa: E402
from nerf_meets_mlx_amd.rendering import ray                          # noqa: E402

# two-sided 97.5 % Student-t quantiles by degrees of freedom (no scipy dependency on the box)
_T975 = {1: 12.706, 2: 4.303, 3: 3.182, 4: 2.776, 5: 2.571, 6: 2.447, 7: 2.365, 8: 2.306, 9: 2.262, 10: 2.228, 11: 2.201,
         12: 2.179, 13: 2.160, 14: 2.145, 15: 2.131, 16: 2.120, 17: 2.110, 18: 2.101, 19: 2.093, 20: 2.086, 21: 2.080,
         22: 2.074, 23: 2.069, 24: 2.064, 25: 2.060, 26: 2.056, 27: 2.052, 28: 2.048, 29: 2.045, 30: 2.042, 31: 2.040,
         35: 2.030, 40: 2.021, 47: 2.012, 50: 2.009, 63: 1.998, 80: 1.990, 100: 1.984}


def t975(df: int) -> float:
    if df <= 0:
        return float("nan")
    ks = sorted(_T975)
    best = max(k for k in ks if k <= df) if df >= ks[0] else ks[0]
    return _T975[best] if df < 200 else 1.96


def paired_stats(d):
    d = np.asarray(d, dtype=np.float64)
    n = len(d)
    if n == 0:
        return {"n": 0}
    mean = float(d.mean())
    sd = float(d.std(ddof=1)) if n > 1 else float("nan")
    half = t975(n - 1) * sd / math.sqrt(n) if n > 1 else float("nan")
    return {"n": n, "mean_delta_db": mean, "std_delta_db": sd, "ci95_half_width_db": half,
            "ci95": [mean - half, mean + half], "median_delta_db": float(np.median(d)),
            "n_positive": int((d > 0).sum()), "n_negative": int((d < 0).sum()),
            "max_abs_delta_db": float(np.abs(d).max())}


def alive_seeds(count: int, quirks: bool, start: int = 0):
    """Seeds whose coarse (seed) AND fine (seed + 1) networks start with sigma > 0: probed with the oracle
    on 512 points, like tools/psnr_parity.alive_seed."""
    from oracle import nerf_oracle as O
    arch = O.NerfArch()
    probe_pos = (torch.rand(64, 8, 3, generator=torch.Generator().manual_seed(0)) - 0.5) * 3.0
    probe_dir = torch.nn.functional.normalize(torch.randn(64, 3, generator=torch.Generator().manual_seed(1)), dim=-1)
    alive = {}

    def is_alive(sd):
        if sd not in alive:
            raw = O.run_model(arch, O.init_params(arch, sd), probe_pos, probe_dir, ref_quirks=quirks)
            alive[sd] = float((raw[..., 3] > 0).float().mean()) > 0.95
        return alive[sd]
    out, sd = [], start
    while len(out) < count:
        if is_alive(sd) and is_alive(sd + 1):
            out.append(sd)
            sd += 2
        else:
            sd += 1
    return out


class DeadTracker:
    """Per arm: at which iterations the coarse / fine loss EQUALS the loss of an empty volume (alpha == 0 everywhere).
    Coarse composites onto white (rgb = 1), the fine loss of the reference composites WITHOUT the white background (Q8:
    rgb = 0) in quirk mode."""

    def __init__(self):
        self.dead = {"coarse": [], "fine": []}

    def update(self, it, losses, target, quirks):
        t = target.double()
        empty = {"coarse": float(((1.0 - t) ** 2).mean()), "fine": float((t ** 2).mean() if quirks else ((1.0 - t) ** 2).mean())}
        for k in ("coarse", "fine"):
            key = "loss_" + k
            if key in losses:
                v = float(losses[key])
                if math.isfinite(v) and abs(v - empty[k]) <= 2e-6 * max(empty[k], 1e-12):
                    self.dead[k].append(it)

    def summary(self, iters):
        out = {}
        for k, v in self.dead.items():
            out[k] = {"dead_iterations": len(v), "first_dead": v[0] if v else None, "last_dead": v[-1] if v else None,
                      "dead_at_end": bool(v) and v[-1] == iters}
        return out


def run_seed(seed, a, dev="cuda", bridge=False, emit=print):
    H = W = a.hw
    imgs, poses, _, _, K = synthetic.make_dataset(H, W, a.views + a.test_views, seed=0, device=dev)
    test_imgs, test_poses = imgs[-a.test_views:], poses[-a.test_views:]
    quirks = not a.no_quirks
    mk = lambda prec: Trainer(imgs[:-a.test_views], poses[:-a.test_views], K, N_rand=a.n_rand, n_depth_samples=64,
                              N_importance=a.n_importance, seed=seed, device=dev, lrate_decay=a.lrate_decay,
                              ref_quirks=quirks, precision=prec)
    arms = {"bf16": mk(getattr(a, "lead_precision", 16)), "fp32": mk(32)}        # key "bf16" = the lead arm (renamed on output when it is not 16)
    assert torch.equal(arms["bf16"].coarse.params, arms["fp32"].coarse.params)
    if getattr(a, "lead_only", False):
        # only the lead arm: its batches, uniforms, initial weights and evaluation pixels are functions of (seed, iteration), so the
        # rows pair with the fp32 / bf16 arms of an EARLIER run of the same configuration and seeds (tools/psnr_converged_stats.py
        # merges rows by (seed, iter))
        del arms["fp32"]
    if getattr(a, "null_arm", False):
        # NULL arm: the SAME bf16 arithmetic, only the fp32 summation ORDER of the weight-gradient reduction differs (the
        # dW kernel's split-K count, an A/B knob: 240 workgroups instead of one per CU) -- relative differences of 1e-7 per
        # gradient.  delta(bf16b - bf16) is what "two runs of one arithmetic" look like under this training recipe: the
        # yardstick for the bf16 - fp32 differences.
        arms["bf16b"] = mk(getattr(a, "lead_precision", 16))
    dead = {k: DeadTracker() for k in arms}
    ot = None
    if bridge:
        from oracle import nerf_oracle as O
        torch.backends.cuda.matmul.allow_tf32 = False
        try:
            torch.set_float32_matmul_precision("highest")
        except Exception:
            pass
        arch = O.NerfArch()
        ot = O.OracleTrainer(arch, 64, a.n_importance, seed=seed, lrate_decay=a.lrate_decay, ref_quirks=quirks, device=dev)
        dead["oracle"] = DeadTracker()
    idx = torch.arange(H * W, device=dev, dtype=torch.int64)
    if 0 < getattr(a, "eval_pixels", 0) < H * W:
        # held-out PSNR on a FIXED pixel subset of every test view (the same for every seed, arm and checkpoint): an
        # unbiased estimate of the view's MSE that keeps the fp32 arm's evaluations from dominating a 20 000-iteration run
        idx = torch.randperm(H * W, generator=torch.Generator().manual_seed(11))[:a.eval_pixels].sort().values.to(dev)
    test_rays = [ray.gen_rays(H, W, K, p[:3, :4].cpu().numpy(), 2.0, 6.0, idx) for p in test_poses]
    test_imgs = [img.reshape(-1, 3)[idx] for img in test_imgs]
    NI = max(a.n_importance, 1)
    u_eval = torch.rand(idx.numel(), NI, generator=torch.Generator().manual_seed(7)).to(dev)

    def psnr_of(tr):
        vals = []
        for img, rays in zip(test_imgs, test_rays):
            rgb = tr.render_rays(rays, u=u_eval if a.n_importance > 0 else None)
            vals.append(float(10.0 * torch.log10(1.0 / torch.mean((rgb - img.reshape(-1, 3)) ** 2))))
        return float(np.mean(vals))

    def psnr_oracle():
        from oracle import nerf_oracle as O
        with torch.no_grad():
            pc = O.unflatten_params(arch, ot.pc.detach())
            pf = O.unflatten_params(arch, ot.pf.detach()) if ot.pf is not None else None
            vals = []
            for img, rays in zip(test_imgs, test_rays):
                outs = []
                for s in range(0, rays.shape[0], 8192):
                    r = rays[s:s + 8192]
                    o = (O.render_rays_eval(arch, pc, pf, r, 64, a.n_importance, u_eval[s:s + 8192], True, False, quirks)
                         if a.n_importance > 0 else O.render_rays(arch, pc, r, 64, True, ref_quirks=quirks))
                    outs.append(o["rgb_map"])
                vals.append(float(O.psnr(torch.cat(outs, 0), img.reshape(-1, 3))))
        return float(np.mean(vals))

    lead = arms["bf16"]
    iters = a.bridge_iters if bridge and a.bridge_iters else a.iters
    recs = []
    t0 = time.time()
    for it in range(1, iters + 1):
        rays, target = lead.sample_batch()                       # (seed, rank, it) -> identical for both arms anyway
        u = lead.train_uniforms(rays.shape[0]) if a.n_importance > 0 else None
        for name, tr in arms.items():
            if name == "bf16b":
                _native.check(_native.lib().nerf_set_option(b"dw_workgroups", 240))
            losses = tr.train_step(rays, target, u)
            if name == "bf16b":
                _native.check(_native.lib().nerf_set_option(b"dw_workgroups", 0))
            if it % a.dead_every == 0 or it == iters:
                dead[name].update(it, losses, target, quirks)
        if ot is not None:
            lo = ot.step(rays[:, 0:3], rays[:, 3:6], target, u)
            if it % a.dead_every == 0 or it == iters:
                dead["oracle"].update(it, lo, target, quirks)
        if it % a.every == 0 or it == iters:
            rec = {"seed": seed, "iter": it, "psnr_bf16": psnr_of(arms["bf16"]), "elapsed_s": round(time.time() - t0, 1)}
            if "fp32" in arms:
                rec["psnr_fp32"] = psnr_of(arms["fp32"])
                rec["delta_db"] = rec["psnr_bf16"] - rec["psnr_fp32"]
            if "bf16b" in arms:
                rec["psnr_bf16b"] = psnr_of(arms["bf16b"])
                rec["delta_db_null"] = rec["psnr_bf16b"] - rec["psnr_bf16"]
            if ot is not None:
                rec["psnr_oracle"] = psnr_oracle()
                rec["delta_db_fp32_minus_oracle"] = rec["psnr_fp32"] - rec["psnr_oracle"]
                rec["delta_db_bf16_minus_oracle"] = rec["psnr_bf16"] - rec["psnr_oracle"]
            recs.append(rec)
            emit(json.dumps(rec))
            if getattr(a, "resync", False):
                # short-horizon mode: the bf16 arm restarts from the fp32 arm's state (weights, Adam moments and step
                # counts), so that every checkpoint measures the divergence accumulated over ONE interval of `every`
                # iterations from a common state -- common random numbers + resynchronisation: trajectories stay
                # correlated over a short horizon, and a systematic per-interval drift shows with a tiny variance
                arms["bf16"].load_state_dict(arms["fp32"].state_dict())
    emit(json.dumps({"seed": seed, "dead_sigma": {k: v.summary(iters) for k, v in dead.items()}, "iters": iters,
                     "bridge": bool(bridge)}))
    return recs, {k: v.summary(iters) for k, v in dead.items()}


def summarise(all_recs, key_a="psnr_bf16", key_b="psnr_fp32", label="bf16_minus_fp32", only_seeds=None):
    """Paired statistics per checkpoint.  only_seeds: restrict to these seeds (e.g. those whose networks are alive at the
    end in BOTH arms: a seed that is dead in both contributes an exact 0 and would pull the mean towards 0)."""
    table = {}
    for r in all_recs:
        if key_a in r and key_b in r and (only_seeds is None or r["seed"] in only_seeds):
            table.setdefault(r["iter"], []).append((r["seed"], r[key_a], r[key_b]))
    out = []
    for it, rows in sorted(table.items()):
        ok = [(s, x, y) for s, x, y in rows if math.isfinite(x) and math.isfinite(y)]
        bad = [s for s, x, y in rows if not (math.isfinite(x) and math.isfinite(y))]
        st = paired_stats([x - y for _, x, y in ok])
        st.update({"ensemble_iter": it, "pair": label, "seeds_used": [s for s, _, _ in ok], "seeds_non_finite": bad,
                   "mean_a": float(np.mean([x for _, x, _ in ok])) if ok else None,
                   "mean_b": float(np.mean([y for _, _, y in ok])) if ok else None,
                   "std_a": float(np.std([x for _, x, _ in ok], ddof=1)) if len(ok) > 1 else None,
                   "std_b": float(np.std([y for _, _, y in ok], ddof=1)) if len(ok) > 1 else None})
        out.append(st)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=32)
    ap.add_argument("--seed-start", type=int, default=0)
    ap.add_argument("--seed-list", default="", help="explicit comma list of seeds (skips the alive-seed scan): lets one ensemble be split over several calls")
    ap.add_argument("--iters", type=int, default=2500)
    ap.add_argument("--every", type=int, default=250)
    ap.add_argument("--hw", type=int, default=100)
    ap.add_argument("--n-rand", type=int, default=1024)
    ap.add_argument("--views", type=int, default=12)
    ap.add_argument("--test-views", type=int, default=2)
    ap.add_argument("--n-importance", type=int, default=128)
    ap.add_argument("--lrate-decay", type=int, default=500)
    ap.add_argument("--dead-every", type=int, default=10, help="check the dead-sigma state every this many iterations (a host sync each)")
    ap.add_argument("--bridge", type=int, default=0, help="first K seeds: also run the fp32 oracle trainer in lockstep")
    ap.add_argument("--bridge-iters", type=int, default=0, help="iterations of the bridge runs (0: --iters)")
    ap.add_argument("--bridge-only", action="store_true")
    ap.add_argument("--no-quirks", action="store_true")
    ap.add_argument("--null-arm", action="store_true", help="third arm: bf16 again with another fp32 summation order in dW (the "
                    "noise floor of this training recipe: what two runs of ONE arithmetic look like)")
    ap.add_argument("--resync", action="store_true", help="after every checkpoint copy the fp32 arm's state into the bf16 arm: each "
                    "checkpoint's delta is then the drift of ONE interval from a common state (short-horizon bias estimator)")
    ap.add_argument("--eval-pixels", type=int, default=0, help="evaluate the held-out PSNR on this many fixed pixels per test view (0: all)")
    ap.add_argument("--out", default="", help="also append every line to this file")
    ap.add_argument("--lead-only", action="store_true", help="train the lead arm only (pairs with the arms of an earlier run of the same seeds)")
    ap.add_argument("--lead-precision", type=int, default=16, choices=[16, 22], help="precision of the lead arm (and of the null arm): 16 = "
                    "bf16 operands; 22 = the float32-tolerance mode on the 16-bit matrix pipe (split-bf16 training, split-fp16 rendering); "
                    "with 22 every 'bf16' in the output keys reads 'p22'")
    a = ap.parse_args()
    fp = open(a.out, "a") if a.out else None

    def emit(line):
        if a.lead_precision != 16:
            line = line.replace("bf16", f"p{a.lead_precision}")
        print(line, flush=True)
        if fp:
            fp.write(line + "\n"); fp.flush()
    seeds = [int(x) for x in a.seed_list.split(",") if x] if a.seed_list else alive_seeds(a.seeds, not a.no_quirks, a.seed_start)
    emit(json.dumps({"config": vars(a), "seeds": seeds, "arms": {"bf16": f"Trainer(precision={a.lead_precision})", "fp32": "Trainer(precision=32)"}}))
    if a.bridge > 0:
        brecs = []
        for sd in seeds[:a.bridge]:
            r, _ = run_seed(sd, a, bridge=True, emit=emit)
            brecs += r
        for st in summarise(brecs, "psnr_fp32", "psnr_oracle", "bridge_fp32_minus_oracle"):
            emit(json.dumps(st))
        for st in summarise(brecs, "psnr_bf16", "psnr_oracle", "bridge_bf16_minus_oracle"):
            emit(json.dumps(st))
    if not a.bridge_only:
        recs, deads = [], {}
        for sd in seeds:
            r, dsum = run_seed(sd, a, bridge=False, emit=emit)
            recs += r
            deads[sd] = dsum
        for st in summarise(recs):
            emit(json.dumps(st))
        if a.null_arm:
            for st in summarise(recs, "psnr_bf16b", "psnr_bf16", "null_bf16b_minus_bf16"):
                emit(json.dumps(st))
        if a.lead_only:
            return
        dead_any = lambda d: any(d[arm][net]["dead_at_end"] for arm in ("bf16", "fp32") for net in ("coarse", "fine"))
        alive = {sd for sd, d in deads.items() if not dead_any(d)}
        for st in summarise(recs, label="bf16_minus_fp32_alive_at_end_in_both_arms", only_seeds=alive):
            emit(json.dumps(st))
        emit(json.dumps({"dead_sigma_summary": {
            arm: {"seeds_ever_dead_fine": [s for s, d in deads.items() if d[arm]["fine"]["dead_iterations"] > 0],
                  "seeds_dead_at_end_fine": [s for s, d in deads.items() if d[arm]["fine"]["dead_at_end"]],
                  "seeds_ever_dead_coarse": [s for s, d in deads.items() if d[arm]["coarse"]["dead_iterations"] > 0],
                  "seeds_dead_at_end_coarse": [s for s, d in deads.items() if d[arm]["coarse"]["dead_at_end"]]}
            for arm in ("bf16", "fp32")}}))


if __name__ == "__main__":
    main()
