"""The statistics behind DESIGN.md 5.3 (f) / (g) (tools/psnr_converged_stats.py: Kaplan-Meier, log-rank, exact sign test) against
scipy's implementations on random censored samples -- the tool itself uses numpy only (it also runs where scipy is absent)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import psnr_converged_stats as S                                    # noqa: E402

stats = pytest.importorskip("scipy.stats")


def _sample(rng, n, lo, p_event):
    return {i: (int(t) * 1000, bool(e)) for i, (t, e) in enumerate(zip(rng.integers(lo, 21, n), rng.random(n) < p_event))}


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_logrank_and_kaplan_meier_match_scipy(seed):
    if not hasattr(stats, "logrank"):
        pytest.skip("scipy without stats.logrank")
    rng = np.random.default_rng(seed)
    ta, tb = _sample(rng, 24, 1, 0.7), _sample(rng, 24, 3 + seed, 0.6)
    lr = S.logrank(ta, tb)
    cd = lambda tt: stats.CensoredData(uncensored=[t for t, e in tt.values() if e], right=[t for t, e in tt.values() if not e])
    ref = stats.logrank(cd(ta), cd(tb))
    assert abs(lr["chi2"] - ref.statistic ** 2) <= 1e-9 * max(1.0, ref.statistic ** 2)
    assert abs(lr["p"] - ref.pvalue) <= 1e-9
    curve, median = S.kaplan_meier(ta)
    sf = stats.ecdf(cd(ta)).sf
    ours = {c["iter"]: c["frac_below_threshold"] for c in curve}
    for q, pr in zip(sf.quantiles, sf.probabilities):
        if int(q) in ours:
            assert abs(ours[int(q)] - pr) <= 1e-12
    if median is not None:
        assert ours[median] <= 0.5 and all(v > 0.5 for k, v in ours.items() if k < median)


def test_sign_test_is_the_exact_two_sided_binomial():
    for a, b in ((6, 5), (10, 7), (0, 5), (3, 3), (12, 1)):
        assert abs(S.sign_test_p(a, b) - stats.binomtest(min(a, b), a + b, 0.5).pvalue) <= 1e-12
    assert S.sign_test_p(0, 0) == 1.0


def test_rows_of_later_files_add_arms_to_a_seed_iteration_row(tmp_path):
    a, b = tmp_path / "a.jsonl", tmp_path / "b.jsonl"
    a.write_text('{"seed": 4, "iter": 1000, "psnr_bf16": 20.0, "psnr_fp32": 21.0}\n{"config": {}}\n')
    b.write_text('{"seed": 4, "iter": 1000, "psnr_p22": 20.5}\n')
    rows = S.load_rows([str(a), str(b)])
    assert rows[(4, 1000)]["psnr_p22"] == 20.5 and rows[(4, 1000)]["psnr_fp32"] == 21.0
    out = S.analyse(rows, ["p22", "fp32"], 20.2)
    tt = {r["arm"]: r for r in out if r["stat"] == "iterations_to_threshold"}
    assert tt["p22"]["reached"] == 1 and tt["fp32"]["reached"] == 1
