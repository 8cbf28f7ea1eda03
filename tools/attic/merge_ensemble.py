#!/usr/bin/env python3
"""Merge the per-call outputs of tools/psnr_ensemble.py (one ensemble split over several gpurun calls by --seed-list) into
one tracked file: all per-(seed, checkpoint) rows, the per-seed dead-sigma records, and the ensemble statistics recomputed
over ALL seeds (paired mean / std / 95 % CI of PSNR_bf16 - PSNR_fp32 per checkpoint; all finite seeds, and the subset whose
networks are alive at the end in both arms).

    python tools/merge_ensemble.py profiles/r03_psnr_ensemble_X.jsonl gpurun_out/r3_ens_A.jsonl gpurun_out/r3_ens_B.jsonl ...
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from psnr_ensemble import paired_stats, summarise                     # noqa: E402  (torch import only; no GPU needed)


def main():
    if len(sys.argv) < 3 or sys.argv[1].startswith("-"):
        print(__doc__)
        return 2
    out, ins = sys.argv[1], sys.argv[2:]
    rows, deads, cfg, seeds, arms, lead = [], {}, None, [], None, 16
    for p in ins:
        for ln in open(p):
            # a lead arm other than bf16 (psnr_ensemble.py --lead-precision 22) is written with 'p22' in place of 'bf16':
            # merged under the tool's internal key and renamed again on output
            if '"lead_precision": 22' in ln:
                lead = 22
            ln = ln.replace("p22", "bf16") if lead == 22 else ln
            r = json.loads(ln)
            if "config" in r:
                cfg = cfg or r["config"]
                arms = arms or r.get("arms")
                seeds += r["seeds"]
            elif "dead_sigma" in r and not r.get("bridge"):
                deads[r["seed"]] = r["dead_sigma"]
            elif "iter" in r and "seed" in r and "psnr_oracle" not in r:
                rows.append(r)
    done = sorted({r["seed"] for r in rows})
    class _Out:
        def __init__(self, f):
            self.f = f
        def write(self, t):
            self.f.write(t.replace("bf16", "p22") if lead == 22 else t)
    with open(out, "w") as raw_fp:
        fp = _Out(raw_fp)
        keep = {k: cfg[k] for k in ("hw", "n_rand", "iters", "every", "views", "test_views", "n_importance", "lrate_decay", "no_quirks")}
        fp.write(json.dumps({"merged_from": [os.path.basename(p) for p in ins], "config": keep, "seeds": done,
                             "arms": arms or {"bf16": "Trainer(precision=16)", "fp32": "Trainer(precision=32)"}}) + "\n")
        for r in sorted(rows, key=lambda r: (r["seed"], r["iter"])):
            fp.write(json.dumps(r) + "\n")
        for sd in done:
            if sd in deads:
                fp.write(json.dumps({"seed": sd, "dead_sigma": deads[sd]}) + "\n")
        for st in summarise(rows):
            fp.write(json.dumps(st) + "\n")
        if any("psnr_bf16b" in r for r in rows):
            for st in summarise(rows, "psnr_bf16b", "psnr_bf16", "null_bf16b_minus_bf16"):
                fp.write(json.dumps(st) + "\n")
            for st in summarise([r for r in rows if "psnr_bf16b" in r], label="bf16_minus_fp32_on_the_null_arm_seeds"):
                fp.write(json.dumps(st) + "\n")
        dead_any = lambda d: any(d[arm][net]["dead_at_end"] for arm in ("bf16", "fp32") for net in ("coarse", "fine"))
        alive = {sd for sd in done if sd in deads and not dead_any(deads[sd])}
        for st in summarise(rows, label="bf16_minus_fp32_alive_at_end_in_both_arms", only_seeds=alive):
            fp.write(json.dumps(st) + "\n")
        # robust location of the heavy-tailed paired differences: median and 10 %-trimmed mean with bootstrap 95 % CIs
        import numpy as np
        rng = np.random.default_rng(0)
        tm = lambda x: float(np.mean(np.sort(x)[int(0.1 * len(x)):len(x) - int(0.1 * len(x))]))
        for it in sorted({r["iter"] for r in rows}):
            d = np.array([r["delta_db"] for r in rows if r["iter"] == it and np.isfinite(r.get("delta_db", float("nan")))])
            if len(d) < 8:
                continue
            bm = np.array([np.median(rng.choice(d, len(d))) for _ in range(4000)])
            bt = np.array([tm(rng.choice(d, len(d))) for _ in range(4000)])
            fp.write(json.dumps({"robust_iter": it, "n": int(len(d)), "median_delta_db": float(np.median(d)),
                                 "median_ci95_bootstrap": [float(x) for x in np.percentile(bm, [2.5, 97.5])],
                                 "trimmed10_mean_delta_db": tm(d), "trimmed10_ci95_bootstrap": [float(x) for x in np.percentile(bt, [2.5, 97.5])],
                                 "n_within_0.5_db": int((np.abs(d) <= 0.5).sum()), "n_beyond_2_db": int((np.abs(d) > 2).sum()),
                                 "min_delta_db": float(d.min()), "max_delta_db": float(d.max())}) + "\n")
        # when each arm first leaves / enters the dead-sigma state
        fp.write(json.dumps({"dead_sigma_summary": {
            arm: {f"seeds_ever_dead_{net}": [s for s in done if s in deads and deads[s][arm][net]["dead_iterations"] > 0] for net in ("coarse", "fine")}
            | {f"seeds_dead_at_end_{net}": [s for s in done if s in deads and deads[s][arm][net]["dead_at_end"]] for net in ("coarse", "fine")}
            for arm in ("bf16", "fp32")}}) + "\n")
    print(f"{len(done)} seeds, {len(rows)} rows -> {out}")
    if any("psnr_bf16b" in r for r in rows):
        for st in summarise(rows, "psnr_bf16b", "psnr_bf16", "null"):
            print("NULL", st["ensemble_iter"], "n", st["n"], "mean %+.3f sd %.3f ci +-%.3f median %+.3f  +%d/-%d" %
                  (st["mean_delta_db"], st["std_delta_db"], st["ci95_half_width_db"], st["median_delta_db"], st["n_positive"], st["n_negative"]))
    for st in summarise(rows):
        print(st["ensemble_iter"], "n", st["n"], "mean %+.3f sd %.3f ci +-%.3f median %+.3f  +%d/-%d" %
              (st["mean_delta_db"], st["std_delta_db"], st["ci95_half_width_db"], st["median_delta_db"], st["n_positive"], st["n_negative"]))


if __name__ == "__main__":
    main()
