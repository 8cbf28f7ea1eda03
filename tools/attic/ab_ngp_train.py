"""A/B of the configs[4] training step (NGPTrainer, 4096 rays x 64 samples): precision 16 with "dw16_variant" 2 / 1 (round 2's 16-wave
kernel / the wave-private pipelines over the bf16 stores), precision 22 with "dw_private_tiles" 0 / 4; alternating blocks on one trainer."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from nerf_meets_mlx_amd import _native
from nerf_meets_mlx_amd.dataset import synthetic
from nerf_meets_mlx_amd.engine.ngp import NGPTrainer
dev = torch.device("cuda", 0)
imgs, poses, rposes, hwf, K = synthetic.make_dataset(800, 800, 4, seed=0, device=dev)
for prec, key, vals in ((16, b"dw16_variant", (2, 1)), (22, b"dw_private_tiles", (0, 4))):
    tr = NGPTrainer(imgs, poses, K, N_rand=4096, seed=7, device=dev, precision=prec)
    for _ in range(10): tr.train_step()
    res = {v: [] for v in vals}
    for r in range(3):
        for v in vals:
            _native.check(_native.lib().nerf_set_option(key, v))
            for _ in range(3): tr.train_step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40): tr.train_step()
            e1.record(); torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / 40)
    for v in vals:
        print(f"precision {prec} {key.decode()} = {v}: configs[4] training step {np.mean(res[v]):.4f} ms ({4096 / np.mean(res[v]) / 1e3:.3f} M rays/s)", flush=True)
    _native.check(_native.lib().nerf_set_option(key, vals[-1]))
