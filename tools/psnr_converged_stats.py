#!/usr/bin/env python3
"""Statistics of a paired PSNR ensemble carried into the converged regime (round 4).

Training under the reference's formulas spends an unpredictable number of iterations on a ~19-20 dB plateau before it
escapes: at a fixed iteration count the PSNR of an arm is bimodal (still on the plateau / escaped), and a
mean of paired differences is dominated by which arm of a pair happened to escape first.  The estimators here are built
for that:

  (a) per arm, the distribution of ITERATIONS-TO-THRESHOLD (first checkpoint with held-out PSNR >= --threshold, censored
      at the last checkpoint): Kaplan-Meier estimate of the fraction still below the threshold at every checkpoint, KM
      median, and the log-rank test between two arms (1 degree of freedom) + the paired sign test of "who got there
      first" (the seeds ARE paired: identical initial weights and batches);
  (b) the paired delta at the first checkpoint where BOTH arms of a seed are above the threshold, and at the end restricted
      to the seeds where both arms are above it ("converged in both arms");
  (c) final PSNR per arm: mean, standard deviation, Student-t 95 % interval, and the same over the converged seeds.

Input: the .jsonl written by tools/psnr_ensemble.py / tools/merge_ensemble.py (rows with seed, iter, psnr_<arm>).
    python tools/psnr_converged_stats.py profiles/r04_psnr_converged_X.jsonl [--threshold 25] [--arms bf16,fp32,bf16b]
Prints one JSON object per statistic (append it to the tracked file with --append).
"""
import argparse
import json
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _t975(df):
    from psnr_ensemble import t975
    return t975(df)


def load_rows(paths):
    rows = {}
    for p in paths:
        for ln in open(p):
            ln = ln.strip()
            if not ln:
                continue
            r = json.loads(ln)
            if "seed" in r and "iter" in r and any(k.startswith("psnr_") for k in r):
                rows.setdefault((r["seed"], r["iter"]), {}).update(r)      # later files add arms to / replace values of a (seed, iter) row
    return rows


def time_to_threshold(rows, arm, thr):
    """{seed: (iteration of the first checkpoint with PSNR >= thr, True) or (last checkpoint, False = censored)}"""
    out = {}
    for sd in sorted({s for s, _ in rows}):
        its = sorted(it for s, it in rows if s == sd and f"psnr_{arm}" in rows[(s, it)])
        if not its:
            continue
        hit = next((it for it in its if rows[(sd, it)][f"psnr_{arm}"] >= thr), None)
        out[sd] = (hit, True) if hit is not None else (its[-1], False)
    return out


def kaplan_meier(tt):
    """tt: {seed: (time, event)} -> [(time, at_risk, events, survival)] at the event times, and the KM median (None when the
    curve never reaches 0.5)."""
    times = sorted({t for t, e in tt.values() if e})
    s, curve, median = 1.0, [], None
    for t in times:
        at_risk = sum(1 for u, _ in tt.values() if u >= t)
        d = sum(1 for u, e in tt.values() if u == t and e)
        s *= 1.0 - d / at_risk
        curve.append({"iter": t, "at_risk": at_risk, "events": d, "frac_below_threshold": s})
        if median is None and s <= 0.5:
            median = t
    return curve, median


def logrank(tt_a, tt_b):
    """Two-sample log-rank test (Mantel-Haenszel), chi-square with 1 degree of freedom."""
    times = sorted({t for t, e in list(tt_a.values()) + list(tt_b.values()) if e})
    o_a = e_a = var = 0.0
    for t in times:
        na = sum(1 for u, _ in tt_a.values() if u >= t)
        nb = sum(1 for u, _ in tt_b.values() if u >= t)
        da = sum(1 for u, e in tt_a.values() if u == t and e)
        db = sum(1 for u, e in tt_b.values() if u == t and e)
        n, d = na + nb, da + db
        if n < 2 or d == 0:
            continue
        o_a += da
        e_a += d * na / n
        var += d * (na / n) * (1 - na / n) * (n - d) / (n - 1)
    if var <= 0:
        return {"chi2": 0.0, "p": 1.0, "observed_a": o_a, "expected_a": e_a}
    chi2 = (o_a - e_a) ** 2 / var
    return {"chi2": chi2, "p": math.erfc(math.sqrt(chi2 / 2.0)), "observed_a": o_a, "expected_a": e_a}


def sign_test_p(n_pos, n_neg):
    """Two-sided exact binomial test of n_pos vs n_neg at p = 1/2 (ties dropped)."""
    n = n_pos + n_neg
    if n == 0:
        return 1.0
    k = min(n_pos, n_neg)
    tail = sum(math.comb(n, i) for i in range(k + 1)) / 2.0 ** n
    return min(1.0, 2.0 * tail)


def mean_ci(x):
    x = np.asarray(x, dtype=np.float64)
    bad = int((~np.isfinite(x)).sum())                       # a diverged arm (NaN PSNR): counted, not averaged
    x = x[np.isfinite(x)]
    n = len(x)
    if n == 0:
        return {"n": 0, "non_finite": bad}
    sd = float(x.std(ddof=1)) if n > 1 else float("nan")
    half = _t975(n - 1) * sd / math.sqrt(n) if n > 1 else float("nan")
    return {"n": n, "non_finite": bad, "mean": float(x.mean()), "std": sd, "ci95_half_width": half,
            "median": float(np.median(x)), "min": float(x.min()), "max": float(x.max())}


def analyse(rows, arms, thr):
    out = []
    seeds = sorted({s for s, _ in rows})
    last = max(it for _, it in rows)
    tts = {a: time_to_threshold(rows, a, thr) for a in arms}
    for a in arms:
        curve, median = kaplan_meier(tts[a])
        out.append({"stat": "iterations_to_threshold", "arm": a, "threshold_db": thr, "n": len(tts[a]),
                    "reached": sum(1 for _, e in tts[a].values() if e), "censored_at": last,
                    "km_median_iter": median, "km_curve": curve,
                    "per_seed": {str(s): (t if e else None) for s, (t, e) in tts[a].items()}})
    for i, a in enumerate(arms):
        for b in arms[i + 1:]:
            common = [s for s in seeds if s in tts[a] and s in tts[b]]
            lr = logrank({s: tts[a][s] for s in common}, {s: tts[b][s] for s in common})
            # paired: who reaches the threshold first (a censored arm counts as later than any reached one; both censored = tie)
            key = lambda tt, s: tt[s][0] if tt[s][1] else float("inf")
            first_a = sum(1 for s in common if key(tts[a], s) < key(tts[b], s))
            first_b = sum(1 for s in common if key(tts[b], s) < key(tts[a], s))
            out.append({"stat": "time_to_threshold_comparison", "pair": f"{a}_vs_{b}", "threshold_db": thr, "n": len(common),
                        "logrank": lr, f"{a}_first": first_a, f"{b}_first": first_b, "ties": len(common) - first_a - first_b,
                        "sign_test_p": sign_test_p(first_a, first_b)})
            # (b) paired delta where both arms are converged
            d_first, d_end, both_end = [], [], []
            for s in common:
                if tts[a][s][1] and tts[b][s][1]:
                    it = max(tts[a][s][0], tts[b][s][0])
                    its = sorted(t for (q, t) in rows if q == s and t >= it)
                    it_both = next((t for t in its if rows[(s, t)][f"psnr_{a}"] >= thr and rows[(s, t)][f"psnr_{b}"] >= thr), None)
                    if it_both is not None:
                        d_first.append(rows[(s, it_both)][f"psnr_{a}"] - rows[(s, it_both)][f"psnr_{b}"])
                r = rows.get((s, last))
                if r and f"psnr_{a}" in r and f"psnr_{b}" in r:
                    d_end.append(r[f"psnr_{a}"] - r[f"psnr_{b}"])
                    if r[f"psnr_{a}"] >= thr and r[f"psnr_{b}"] >= thr:
                        both_end.append(r[f"psnr_{a}"] - r[f"psnr_{b}"])
            out.append({"stat": "paired_delta_db", "pair": f"{a}_minus_{b}", "threshold_db": thr,
                        "at_first_checkpoint_both_above_threshold": mean_ci(d_first),
                        f"at_iter_{last}_all_seeds": mean_ci(d_end),
                        f"at_iter_{last}_seeds_with_both_arms_above_threshold": mean_ci(both_end)})
    for a in arms:
        fin = [rows[(s, last)][f"psnr_{a}"] for s in seeds if (s, last) in rows and f"psnr_{a}" in rows[(s, last)]]
        conv = [v for v in fin if v >= thr]
        out.append({"stat": "final_psnr_db", "arm": a, "iter": last, "all_seeds": mean_ci(fin),
                    "seeds_above_threshold": mean_ci(conv), "frac_above_threshold": len(conv) / max(1, len(fin))})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="+")
    ap.add_argument("--threshold", type=float, default=25.0)
    ap.add_argument("--arms", default="", help="comma list (default: every psnr_<arm> key found)")
    ap.add_argument("--append", default="", help="append the statistics to this file as JSON lines")
    a = ap.parse_args()
    rows = load_rows(a.files)
    if not rows:
        print("no (seed, iter) rows found", file=sys.stderr)
        return 2
    found = sorted({k[5:] for r in rows.values() for k in r if k.startswith("psnr_")})
    arms = [x for x in a.arms.split(",") if x] or [x for x in ("bf16", "fp32", "bf16b", "f22") if x in found]
    res = analyse(rows, arms, a.threshold)
    fp = open(a.append, "a") if a.append else None
    for r in res:
        line = json.dumps(r)
        print(line)
        if fp:
            fp.write(line + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
