"""world_size-2 gloo tests (CPU) for the N>1 path: rays shard with no data-path collective, the
flat gradient buffer is summed once per network step and scaled by 1/world before Adam.
The compute of each rank is the CPU oracle here (no GPU in this suite); the HIP Trainer uses the same
parallel.* helpers around its kernels (engine/trainer.py:_step_net)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import nerf_oracle as O


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from nerf_meets_mlx_amd import parallel
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and parallel.world() == (rank, world)
    # --- sharding helpers
    lo, hi = parallel.shard_range(640000, rank, world)
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([hi - lo]))
    assert sum(int(s) for s in sizes) == 640000
    seeds = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(seeds, torch.tensor([parallel.rank_seed(7, rank, 3)]))
    assert len({int(s) for s in seeds}) == world
    # --- gradient all-reduce == full-batch gradient (loss is a mean over rays)
    arch = O.NerfArch()
    flat = O.flatten_params(arch, O.init_params(arch, 0))
    g = torch.Generator().manual_seed(1)
    B, n = 8, 4
    o = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1) * 4.0
    d = -o / 4.0 + 0.2 * torch.randn(B, 3, generator=g)
    y = torch.rand(B, 3, generator=g)
    rays = O.pack_rays(o, d, 2.0, 6.0)

    def grads(sl):
        p = flat.clone().requires_grad_(True)
        loss, _ = O.coarse_loss(arch, O.unflatten_params(arch, p), rays[sl], y[sl], n)
        return torch.autograd.grad(loss, p)[0]
    lo, hi = parallel.shard_range(B, rank, world)
    mine = grads(slice(lo, hi))
    parallel.allreduce_sum_(mine)
    mine = mine / world
    full = grads(slice(0, B))
    err = float((mine - full).abs().max() / full.abs().max())
    # --- identical Adam step on every rank
    p1 = flat.clone(); m = torch.zeros_like(p1); v = torch.zeros_like(p1)
    O.adam_step(p1, mine, m, v, 5e-4)
    ps = [torch.zeros_like(p1) for _ in range(world)]
    dist.all_gather(ps, p1)
    same = all(torch.equal(ps[0], t) for t in ps)
    # --- image gather to rank 0
    rows = torch.arange(lo, hi, dtype=torch.float32)[:, None].repeat(1, 3)
    img = parallel.gather_rows_to_rank0(rows, B)
    ok_img = (img is None) if rank else bool(torch.equal(img[:, 0], torch.arange(B, dtype=torch.float32)))
    # --- the communicator describes itself (bench.py prints this in the --gpus N line): every rank answered the all-reduce
    info = parallel.comm_info(allreduce_bytes_per_step=2 * 595844 * 4, collectives_per_step=2)
    ok_info = (info["backend"] == "gloo" and info["world_size"] == world and info["world_size_seen"] == world
               and [d["rank"] for d in info["devices"]] == list(range(world)) and info["allreduce_bytes_per_step"] == 4766752)
    q.put((rank, err, same, ok_img and ok_info))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_world_size_n_gloo(world):
    """world 2, and a rehearsal of the eight ranks of the driver's SCALE run: shard arithmetic, distinct per-rank streams, the summed
    and scaled gradient equals the full-batch gradient, identical Adam steps, the image gather."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, same, ok_img in res:
        assert err < 1e-5, (rank, err)
        assert same and ok_img


def test_single_process_helpers():
    from nerf_meets_mlx_amd import parallel
    assert parallel.world() == (0, 1)
    t = torch.ones(4)
    assert parallel.allreduce_sum_(t) is t
    cover = [parallel.shard_range(10, r, 3) for r in range(3)]
    assert cover == [(0, 3), (3, 6), (6, 10)]
    assert parallel.gather_rows_to_rank0(t[:, None], 4).shape == (4, 1)
    info = parallel.comm_info()
    assert info["backend"] is None and info["world_size_seen"] == 1 and info["devices"] is None
