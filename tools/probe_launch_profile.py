"""Where does a launch of the persistent render-forward kernels spend its time?  (round 6, review item 2: the linear fit over the
two launch sizes of the bench has an intercept of 0.25-0.4 ms -- 3-4 % of the headline -- that no budget explained.)

    make -C nerf_meets_mlx_amd/csrc stamp
    NERF_HIP_LIB=tools/diag/libnerf_stamp.so python tools/probe_launch_profile.py [--context bench|hot]

Diagnostic build only: every workgroup stamps the chip-wide 100 MHz clock (s_memrealtime) at kernel entry, at the start of its
pass loop, after passes 1, 2, 4, ... 128 and at its end, and the shader clock (s_memtime) at four of those points
(csrc/mlp_ring.h, Stamp2).  Printed per launch: the kernel's wall time (events), the spread of workgroup start and end times,
the set-up time before the first pass, the time per pass in each window, the clock in each window, and the same per XCD.
context hot: the measured launch follows 2.5 s of back-to-back launches of itself; bench: it follows what the bench's step puts in
front of it (a training step, coarse sampling / the coarse pass + compositing + importance sampling)."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_meets_mlx_amd import _native, sampling                      # noqa: E402
from nerf_meets_mlx_amd.models.NeRF import NeRF                       # noqa: E402
from nerf_meets_mlx_amd.rendering import render                       # noqa: E402

dev = "cuda"
lib = _native.lib()


def stamps(fn, nwg):
    f = getattr(lib, fn)
    f.restype, f.argtypes = C.c_int, [C.c_void_p, C.c_int]
    st = np.zeros((nwg, 24), dtype=np.uint64)
    assert f(st.ctypes.data, nwg) == 0
    return st.astype(np.float64)


def report(name, ms, st):
    t0 = st[:, 0].min()
    us = lambda a: (a - t0) * 0.01                                 # 100 MHz ticks -> microseconds since the first workgroup's entry
    entry, loop, end = us(st[:, 0]), us(st[:, 1]), us(st[:, 10])
    passes = st[:, 14]
    print(f"== {name}: kernel {ms * 1e3:.0f} us (events) | {len(st)} workgroups x {int(np.median(passes))} passes (min {int(passes.min())} max {int(passes.max())})")
    print(f"   entry spread {entry.max():.1f} us | set-up (entry -> first pass) median {np.median(loop - entry):.1f} max {(loop - entry).max():.1f} us"
          f" | first pass starts by {loop.max():.1f} us")
    print(f"   workgroup end: min {end.min():.0f} median {np.median(end):.0f} max {end.max():.0f} us  (tail = max - median = {end.max() - np.median(end):.0f} us;"
          f" kernel - last end = {ms * 1e3 - end.max():.0f} us)")
    pts = [(0, 1, "loop")] + [(1 << k, 2 + k, f"p{1 << k}") for k in range(8)]
    prev_t, prev_n, row = st[:, 1], 0, []
    for n_, col, _ in pts[1:]:
        ok = passes >= n_
        if not ok.any():
            break
        per = (st[ok, col] - prev_t[ok]) * 0.01 / (n_ - prev_n)
        row.append(f"{prev_n}-{n_}: {np.median(per):.1f} (max {per.max():.1f})")
        prev_t, prev_n = st[:, col].copy(), n_
    ok = passes > prev_n
    if ok.any():
        per = (st[ok, 10] - prev_t[ok]) * 0.01 / (passes[ok] - prev_n)
        row.append(f"{prev_n}-end: {np.median(per):.1f} (max {per.max():.1f})")
    print("   us per pass by window: " + " | ".join(row))
    ghz = lambda m1, m0, r1, r0: np.median((st[:, m1] - st[:, m0]) / np.maximum(st[:, r1] - st[:, r0], 1) * 0.1)
    ck = [f"entry->loop {ghz(12, 11, 1, 0):.2f}", f"loop->p1 {ghz(16, 12, 2, 1):.2f}"]
    if passes.min() >= 8:
        ck.append(f"p1->p8 {ghz(17, 16, 5, 2):.2f}")
    if passes.min() >= 32:
        ck.append(f"p8->p32 {ghz(18, 17, 7, 5):.2f}")
        ck.append(f"p32->end {ghz(13, 18, 10, 7):.2f}")
    print("   clock GHz: " + " | ".join(ck) + f" | whole loop {ghz(13, 12, 10, 1):.2f}")
    xcc = (st[:, 15].astype(np.uint64) >> np.uint64(32)).astype(np.int64) & 7
    print("   per XCD (workgroups, median end us, max end us): " + " ".join(f"[{x}: {int((xcc == x).sum())}, {np.median(end[xcc == x]):.0f}, {end[xcc == x].max():.0f}]"
                                                                               for x in range(8) if (xcc == x).any()))


def step_context(B):
    """The launches as the bench's own step runs them, in steady state: 12 full steps (training step at N_rand 4096 + render chunk:
    coarse pass, compositing, importance sampling, fine pass), then one more step that stops right behind the launch of interest --
    its stamps are then the last ones the kernel wrote."""
    from nerf_meets_mlx_amd.dataset import synthetic
    from nerf_meets_mlx_amd.engine.trainer import Trainer
    from nerf_meets_mlx_amd.rendering import ray
    H = W = 800
    imgs, poses, rposes, hwf, K = synthetic.make_dataset(H, W, 4, seed=0, device=dev)
    ridx = torch.arange(0, B, device=dev, dtype=torch.int64)
    rrays = ray.gen_rays(H, W, K, rposes[40][:3, :4], 2.0, 6.0, ridx)
    for prec, fn in ((22, "nerf_debug_stamps2_mlp22"), (16, "nerf_debug_stamps2_ring16")):
        tr = Trainer(imgs, poses, K, N_rand=4096, n_depth_samples=64, N_importance=128, seed=4, device=dev, chunk=B, precision=prec)

        def render_chunk(stop_after_coarse, timed):
            z = sampling.sample_coarse(rrays, 64)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if stop_after_coarse:
                e0.record(); raw = tr.coarse.query(rrays, z); e1.record()
                return e0, e1
            raw = tr.coarse.query(rrays, z)
            _, _, _, w, _ = render.composite(raw, z, rrays, 0.0, True)
            u = torch.rand(B, 128, device=dev, generator=tr.gen)
            _, zf = sampling.importance_sample(z, w, 128, u=u)
            e0.record(); raw = tr.fine.query(rrays, zf); e1.record()
            render.composite(raw, zf, rrays, 0.0, True, need_weights=False)
            return e0, e1
        for which in ("coarse", "fine"):
            for _ in range(12):
                tr.train_step()
                render_chunk(False, False)
            tr.train_step()
            e0, e1 = render_chunk(which == "coarse", True)
            torch.cuda.synchronize()
            n = 64 if which == "coarse" else 192
            nwg = min(256, (B * n + 127) // 128)
            report(f"precision {prec} {fn.split('_')[-1]} render {which} pass {B} x {n} samples inside the bench's step (steady state)",
                   e0.elapsed_time(e1), stamps(fn, nwg))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--context", choices=["hot", "bench", "step"], default="hot")
    ap.add_argument("--rays", type=int, default=32768)
    a = ap.parse_args()
    torch.manual_seed(0)
    B = a.rays
    if a.context == "step":
        return step_context(B)
    o = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1) * 4
    d = -o / 4 + 0.2 * torch.randn(B, 3, device=dev)
    r = torch.zeros(B, 11, device=dev); r[:, :3] = o; r[:, 3:6] = d; r[:, 6] = 2; r[:, 7] = 6
    r[:, 8:] = d / d.norm(dim=-1, keepdim=True)
    for prec, fn in ((22, "nerf_debug_stamps2_mlp22"), (16, "nerf_debug_stamps2_ring16")):
        m = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=dev, seed=0, precision=prec)
        m.load_flat(m.params * 1.5)
        zs = {n: torch.sort(torch.rand(B, n, device=dev) * 4 + 2, -1).values for n in (64, 192)}
        for n in (64, 192):
            z = zs[n]
            if a.context == "hot":
                t0 = time.time()
                while time.time() - t0 < 2.5:
                    for _ in range(20):
                        m.query(r, z)
                    torch.cuda.synchronize()
            else:
                # what the bench's render chunk puts in front of this launch: n = 64 -> idle sync, then coarse sampling; n = 192 ->
                # the coarse pass, compositing, importance sampling
                for _ in range(3):
                    m.query(r, zs[192])
                torch.cuda.synchronize()
                time.sleep(0.05)
                zc = sampling.sample_coarse(r, 64)
                if n == 192:
                    raw = m.query(r, zc)
                    _, _, _, w, _ = render.composite(raw, zc, r, 0.0, True)
                    u = torch.rand(B, 128, device=dev)
                    _, z = sampling.importance_sample(zc, w, 128, u=u)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); m.query(r, z); e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            nwg = min(256, (B * n + 127) // 128)
            report(f"precision {prec} {fn.split('_')[-1]} {B} x {n} samples, context {a.context}", ms, stamps(fn, nwg))


if __name__ == "__main__":
    main()
