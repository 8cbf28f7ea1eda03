"""Multi-GPU glue: rays shard embarrassingly; the ONLY collective is the sum
all-reduce of the flat gradient buffer (2.38 MB fp32 per network) before Adam.  One process per
GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests)."""
import os
from typing import Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_from_env(backend: str = None) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment; no-op for a single process."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if ws > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        local = int(os.environ.get("NERF_FORCE_DEVICE", local))     # rehearsals: several ranks on one GPU (gloo)
        if backend == "nccl":
            torch.cuda.set_device(local)
            try:      # eager communicator on this rank's GPU (no lazy init inside the first collective)
                dist.init_process_group(backend=backend, rank=rank, world_size=ws, device_id=torch.device("cuda", local))
            except TypeError:
                dist.init_process_group(backend=backend, rank=rank, world_size=ws)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=ws)
    return rank, ws, int(os.environ.get("NERF_FORCE_DEVICE", local))


def barrier():
    """Process-group barrier (RCCL barriers are tied to the rank's device)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def comm_info(device=None, allreduce_bytes_per_step: int = 0, collectives_per_step: int = 0) -> dict:
    """What the communicator of this job IS, as seen from inside it (bench.py prints it in the `--gpus N` line so that a
    scaling record explains itself): backend name, the world size the process group reports, the number of ranks that
    actually answered a sum all-reduce of ones (`world_size_seen`: a collective, every rank calls this), the RCCL version
    torch was built against (`torch.cuda.nccl.version()`; None without it), each rank's device, and the bytes the data path
    all-reduces per step.  One process: no communicator, world_size_seen 1."""
    info = {"backend": None, "world_size": 1, "world_size_seen": 1, "rccl_version": None, "devices": None,
            "allreduce_bytes_per_step": int(allreduce_bytes_per_step), "collectives_per_step": int(collectives_per_step)}
    try:
        v = torch.cuda.nccl.version()
        info["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception:                                      # CPU-only build / no RCCL in this torch
        pass
    if not (dist.is_available() and dist.is_initialized()):
        return info
    ws = dist.get_world_size()
    backend = dist.get_backend()
    on_gpu = backend == "nccl" or (device is not None and torch.device(device).type == "cuda")
    dev = torch.device(device if device is not None else ("cuda" if on_gpu else "cpu"))
    one = torch.ones(1, dtype=torch.float32, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    mine = {"rank": dist.get_rank(), "device": str(dev)}
    if dev.type == "cuda":
        pr = torch.cuda.get_device_properties(dev)
        mine.update(index=dev.index if dev.index is not None else torch.cuda.current_device(), name=pr.name,
                    arch=getattr(pr, "gcnArchName", None))
    every = [None] * ws
    dist.all_gather_object(every, mine)
    info.update(backend="rccl (torch.distributed 'nccl')" if backend == "nccl" else backend, world_size=ws,
                world_size_seen=int(round(float(one[0]))), devices=every)
    return info


def shard_range(n: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of n units for `rank`; slices are disjoint and cover [0, n)."""
    return n * rank // world_size, n * (rank + 1) // world_size


def rank_seed(seed: int, rank: int, stream: int = 0) -> int:
    """Disjoint deterministic RNG streams per rank (pixel / image choice)."""
    return (seed * 1000003 + rank * 7919 + stream * 104729) & ((1 << 63) - 1)


def counter_seed(seed: int, rank: int, stream: int, it: int) -> int:
    """Stateless 63-bit seed for (run seed, rank, stream, iteration): the trainers draw everything random of iteration
    `it` from generators seeded with this, so that a run resumed from a checkpoint at iteration `it` -- on any rank,
    whoever wrote the file -- continues exactly like the uninterrupted one; no RNG state needs saving.
    (splitmix64 finaliser over an odd-multiplier combination of the four counters: ~1 us of host time.)"""
    m = (1 << 64) - 1
    x = (int(seed) * 0x9E3779B97F4A7C15 + int(rank) * 0xBF58476D1CE4E5B9 + int(stream) * 0x94D049BB133111EB
         + int(it) * 0xD6E8FEB86659FD93 + 0x2545F4914F6CDD1D) & m
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & m
    x ^= x >> 27
    x = (x * 0x94D049BB133111EB) & m
    x ^= x >> 31
    return x >> 1


# bench.py: a list here receives one (start, end) torch.cuda.Event pair per gradient all-reduce, recorded on the stream the
# collective is issued on (the trainers' comm stream): comm_ms_per_step of the scaling runs.  None = no timing.
comm_timing = None


def allreduce_sum_(flat: torch.Tensor) -> torch.Tensor:
    """In-place sum over ranks of the flat gradient buffer; identity for one process."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if comm_timing is not None and flat.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            e1.record()
            comm_timing.append((e0, e1))
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def gather_rows_to_rank0(local: torch.Tensor, total_rows: int):
    """Concatenate per-rank row slices (shard_range order) on rank 0; returns None elsewhere."""
    rank, ws = world()
    if ws == 1:
        return local
    sizes = [shard_range(total_rows, r, ws) for r in range(ws)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(out, pad)
    if rank != 0:
        return None
    return torch.cat([o[:hi - lo] for o, (lo, hi) in zip(out, sizes)], 0)


class NativeComm:
    """RCCL communicator owned by libnerf_hip (`nerf_comm_*`, `nerf_allreduce_grads`): the same all-reduce without
    going through torch.distributed's process group.  The 128-byte id is created on rank 0 and handed to the other
    ranks by the caller (here: a torch.distributed broadcast, any out-of-band channel works)."""

    def __init__(self, rank: int, world_size: int, device="cuda"):
        import ctypes as C
        from . import _native as N
        self._N, self._C = N, C
        ident = C.create_string_buffer(128)
        if rank == 0:
            N.check(N.lib().nerf_comm_unique_id(ident))
        if world_size > 1:
            t = torch.tensor(list(ident.raw), dtype=torch.uint8, device=device if dist.get_backend() == "nccl" else "cpu")
            dist.broadcast(t, src=0)
            ident = C.create_string_buffer(bytes(t.cpu().tolist()), 128)
        self.comm = C.c_void_p()
        N.check(N.lib().nerf_comm_init(C.byref(self.comm), world_size, rank, ident))

    def allreduce_sum_(self, flat: torch.Tensor) -> torch.Tensor:
        N = self._N
        N.check(N.lib().nerf_allreduce_grads(self.comm, N.ptr(flat), flat.numel(), N.stream()))
        return flat

    def close(self):
        if self.comm:
            self._N.check(self._N.lib().nerf_comm_destroy(self.comm))
            self.comm = self._C.c_void_p()
