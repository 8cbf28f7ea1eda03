"""Probe: hipGraph capture (torch.cuda.CUDAGraph) of one training iteration at small batch sizes."""
import sys, time, torch
sys.path.insert(0, ".")
from nerf_meets_mlx_amd.dataset import synthetic
from nerf_meets_mlx_amd.engine.trainer import Trainer
dev = "cuda"
H = W = 800
imgs, poses, rposes, hwf, K = synthetic.make_dataset(H, W, 2, seed=0, device=dev)
for N in (1024, 4096):
    tr = Trainer(imgs, poses, K, N_rand=N, seed=4, device=dev)
    rays, target = tr.sample_batch()
    u = torch.rand(N, 128, device=dev)
    srays, starget, su = rays.clone(), target.clone(), u.clone()
    for _ in range(3): tr.train_step(srays, starget, su)            # warm-up: attributes, workspaces, packed images
    torch.cuda.synchronize()
    def eager(it=20):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(it): tr.train_step(srays, starget, su)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
    e = eager()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        tr.train_step(srays, starget, su)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = tr.train_step(srays, starget, su)
    torch.cuda.synchronize()
    def replay(it=20):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(it): g.replay()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
    r = replay()
    print(f"N_rand={N}: eager {e:.3f} ms/step, graph replay {r:.3f} ms/step, loss {float(out['loss_coarse']):.5f}", flush=True)
