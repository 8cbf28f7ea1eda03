"""Round-4 GPU tests.

(1) configs[4]: a table gradient left behind by a public HashNeRF.backward() call (or by a step that raised between the
    scatter and Adam) is NOT added to the next training step's gradient (advisor, round 3).
"""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"


def _ngp(det, groups=4, seed=7, **kw):
    from nerf_meets_mlx_amd.dataset import synthetic
    from nerf_meets_mlx_amd.engine.ngp import NGPTrainer
    imgs, poses, _, _, K = synthetic.make_dataset(24, 24, 3, seed=0, device=DEV)
    return NGPTrainer(imgs, poses, K, N_rand=128, n_depth_samples=64, seed=seed, device=DEV, log2_hashmap_size=14,
                      deterministic=det, level_groups=groups, **kw)


@pytest.mark.parametrize("det", [True, False])
def test_ngp_stale_table_gradient_is_not_added_to_the_next_step(det):
    """NGPTrainer.train_step scatters into the accumulator WITHOUT clearing it when the previous step's Adam left it
    zero.  A public field.backward() in between (default accumulate=False: clears, then leaves ITS gradient behind)
    must not leak into the step: parameters after (backward; train_step) == parameters after (train_step) alone."""
    from nerf_meets_mlx_amd import sampling
    from nerf_meets_mlx_amd.rendering import render
    a, b = _ngp(det), _ngp(det)
    for tr in (a, b):
        tr.train_step()                                  # one ordinary step first: Adam state exists, accumulator cleared
    assert a.field._grad_clean and b.field._grad_clean
    rays, target = a.sample_batch()
    rb, tb = b.sample_batch()
    assert torch.equal(rays, rb) and torch.equal(target, tb)
    # a: three stray gradient evaluations on another batch (what tests/test_gpu_parity.py does before train_step)
    other = torch.roll(rays, 7, 0)
    z = sampling.sample_coarse(other, 64)
    for _ in range(3):
        raw = a.field.query(other, z, train=True)
        _, d_raw, _ = render.composite_mse_backward(raw, z, other, torch.roll(target, 3, 0), True)
        a.field.backward(d_raw)
    assert not a.field._grad_clean
    assert float(a.field.table_grad().abs().max()) > 0    # a stale gradient IS sitting in the accumulator
    a.train_step(rays, target)
    b.train_step(rays, target)
    torch.cuda.synchronize()
    assert a.field._grad_clean
    if det:
        assert torch.equal(a.field.enc.tables, b.field.enc.tables)
    else:                                                # float atomics: order-dependent rounding, nothing more
        assert float((a.field.enc.tables - b.field.enc.tables).abs().max()) <= 1e-6 * float(b.field.enc.tables.abs().max()) + 1e-9
    assert torch.equal(a.field.mlp.params, b.field.mlp.params) or not det
    assert float(a.field.table_grad().abs().max()) == 0   # consumed and cleared
