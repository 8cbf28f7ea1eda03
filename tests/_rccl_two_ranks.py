"""Worker of tests/test_gpu_round2.py::test_rccl_two_ranks_allreduce_and_training (launched by torch.distributed.run,
one rank per GPU, backend "nccl" = RCCL).  Not collected by pytest (leading underscore)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from nerf_meets_mlx_amd import parallel                                # noqa: E402
from nerf_meets_mlx_amd.dataset import synthetic                       # noqa: E402
from nerf_meets_mlx_amd.engine.trainer import Trainer                  # noqa: E402


def main():
    rank, world, local = parallel.init_from_env("nccl")
    assert world == 2 and dist.get_backend() == "nccl"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    # (1) the gradient all-reduce through torch.distributed (what the Trainer uses)
    g = torch.full((595844,), float(rank + 1), device=dev)
    parallel.allreduce_sum_(g)
    assert bool((g == 3.0).all()), "torch.distributed RCCL all-reduce"
    # (2) the same through libnerf_hip's own communicator (nerf_comm_* / nerf_allreduce_grads)
    comm = parallel.NativeComm(rank, world, device=dev)
    h = torch.arange(595844, device=dev, dtype=torch.float32) * (rank + 1)
    comm.allreduce_sum_(h)
    torch.cuda.synchronize()
    assert torch.equal(h, torch.arange(595844, device=dev, dtype=torch.float32) * 3), "NativeComm all-reduce"
    comm.close()
    # (3) two training iterations: different rays per rank, identical weights afterwards
    imgs, poses, _, _, K = synthetic.make_dataset(16, 16, 3, seed=0, device=dev)
    tr = Trainer(imgs, poses, K, N_rand=64, n_depth_samples=64, N_importance=128, seed=4, device=dev)
    rays0, _ = tr.sample_batch()
    for _ in range(2):
        tr.train_step()
    ps = [torch.zeros_like(tr.coarse.params) for _ in range(world)]
    dist.all_gather(ps, tr.coarse.params)
    pf = [torch.zeros_like(tr.fine.params) for _ in range(world)]
    dist.all_gather(pf, tr.fine.params)
    rs = [torch.zeros_like(rays0) for _ in range(world)]
    dist.all_gather(rs, rays0)
    assert torch.equal(ps[0], ps[1]) and torch.equal(pf[0], pf[1]), "weights differ between ranks"
    assert not torch.equal(rs[0], rs[1]), "ranks drew the same rays"
    frame = tr.render_frame(poses[0])
    assert (frame is None) if rank else tuple(frame.shape) == (16, 16, 3)
    parallel.barrier()
    if rank == 0:
        print("RCCL2 OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
