#!/usr/bin/env python3
"""bench.py -- train+render rays/s of the MI355X-native NeRF hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: under torch.distributed.run, or alone -- it then starts
                                                             its N ranks itself before touching the GPU)

One "step" = one training iteration of the reference loop (entrypoints/__test_nerf.py:200-305:
N_rand rays, coarse 64 + fine 64+128 samples, forward + backward + Adam for both networks) PLUS
one render chunk (rendering/render.py:243-266: `chunk` = 32768 rays through coarse + importance
sampling + fine) on the synthetic Lego-like 800x800 scene (BASELINE.json configs[2]).  Inputs are
resident in HBM before the timed region.  value = (train rays + render rays) of ALL ranks / time;
rays shard across ranks with no data-path collective except the gradient all-reduce (weak scaling).

The headline step runs at `--precision 22`: the reference's float32 TOLERANCE on the 16-bit matrix pipe (every float32 GEMM
operand as a hi + lo pair of 16-bit numbers, three MFMAs per product, fp32 accumulate: csrc/mlp_s16.hip for training,
csrc/mlp22.hip for rendering), held to the same fixture tolerances as the float32 kernels.  The N = 1 line also carries the
same step with literal float32 operands on the fp32 MFMA (`fp32`, `reference_precision_value`) and in the declared
reduced-precision bf16 mode (`bf16`).

The JSON line also carries `roofline` for the dominant kernel (the fused MLP forward of the render
fine pass, MFMA-bound, timed with events on the launch stream inside the timed region) and
`cpu_baseline` (the CPU oracle = op-for-op restatement of the reference, timed on this box's host
cores on a bounded sample of the same workload; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_SAMPLE_FWD = 2 * 593408          # SURVEY 8d / BASELINE.md section 2
BF16_MFMA_PEAK_TFLOPS = 2500.0            # MI355X_MICROARCH.md: dense bf16 MFMA
FP32_MFMA_PEAK_TFLOPS = 157.3             # MI355X_MICROARCH.md: fp32 matrix (v_mfma_f32_32x32x2_f32)
# HBM bytes per sample of a training pass (8 TB/s peak): fragment blocks written once by forward / chain and read once by
# the weight-gradient jobs + sign-bit words (bf16: DESIGN 4.2; split bf16: hi + lo blocks of everything + the re-read encodings;
# fp32: [tile][row][32] float rows)
TRAIN_BYTES = {16: (321 + 356 + 9) * 1024 / 32, 22: (633 + 712 + 9 + 12) * 1024 / 32, 32: (2592 + 2496) * 128 * 2 / 32}
PEAK_TFLOPS = {16: BF16_MFMA_PEAK_TFLOPS, 22: BF16_MFMA_PEAK_TFLOPS / 3.0, 32: FP32_MFMA_PEAK_TFLOPS}
KERNEL = {16: "mlp_fwd_ring16_kernel<8,2>", 22: "mlp22_fwd_kernel<1,3>", 32: "mlp32_fwd_kernel"}
# what the MFMA OPERANDS are (the label says operands, not the tolerance met): precision 22 carries every float32 GEMM operand as
# two 16-bit numbers -- fp16 x 2 in rendering (22 significand bits), bf16 x 2 in training (16 significand bits) -- and accumulates
# in fp32; it is held to the float32 kernels' fixture tolerances (1e-4 of the output scale), but its operands are NOT float32
DTYPE = {16: "bf16", 22: "split16 (fp16x2 render / bf16x2 train MFMA operands ~ 22 / 16 significand bits; fp32 accumulate)", 32: "f32"}
DTYPE_SHORT = {16: "bf16", 22: "split16", 32: "f32"}
OPERAND_BITS = {16: {"render": 8, "train": 8}, 22: {"render": 22, "train": 16}, 32: {"render": 24, "train": 24}}
NGP_DTYPE = {16: "bf16 (MLP operands; fp16 shadow-table gathers, float32 interpolation rounded to bf16)",
             22: "split16 (bf16x2 MLP operands ~ 16 significand bits, fp32 accumulate; float32 table gathers and interpolation)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50, help="timed steps (50 x ~12 ms: long enough for an SMI sampler to see the GPU busy)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n-rand", type=int, default=4096, help="training rays per GPU per step (argparse default of the reference)")
    ap.add_argument("--render-rays", type=int, default=32768, help="rays per render chunk per GPU per step")
    ap.add_argument("--hw", type=int, default=800)
    ap.add_argument("--train-images", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the sustained / fp32 / ngp legs of the default N=1 line")
    ap.add_argument("--sustain-seconds", type=float, default=10.0, help="length of the sustained bf16 leg")
    ap.add_argument("--fp32-steps", type=int, default=20, help="timed steps of the fp32 (reference arithmetic) leg")
    ap.add_argument("--cpu-seconds", type=float, default=30.0, help="time box of the cpu_baseline render leg (the training leg runs "
                    "its 20 steps of SURVEY 8(d) whatever they take, capped at --cpu-train-cap seconds)")
    ap.add_argument("--cpu-train-cap", type=float, default=600.0, help="safety cap of the cpu_baseline training leg")
    ap.add_argument("--precision", type=int, default=22, choices=[16, 22, 32],
                    help="precision of the headline step: 22 = the reference's float32 tolerance on the 16-bit matrix pipe (default), "
                         "16 = bf16 operands (declared reduced precision), 32 = float32 operands on the fp32 MFMA")
    ap.add_argument("--mlp-variant", type=int, default=0)
    ap.add_argument("--n-importance", type=int, default=128, help="0 = coarse-only (BASELINE configs[1] with --hw 400)")
    ap.add_argument("--config", choices=["nerf", "ngp"], default="nerf",
                    help="nerf: 8x256 NeRF, the headline workload (BASELINE configs[1-3]); ngp: hash grid + 2x64 MLP (configs[4])")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Not under a launcher: start one rank per GPU ourselves (torch.distributed.run, RCCL rendezvous on 127.0.0.1)
        # BEFORE anything in this process touches the GPU, and leave with the children's exit code.
        sys.exit(spawn_ranks(args.gpus))

    from nerf_meets_mlx_amd import _native, parallel
    from nerf_meets_mlx_amd.dataset import synthetic
    from nerf_meets_mlx_amd.engine.trainer import Trainer
    from nerf_meets_mlx_amd.rendering import ray

    rank, world, local = parallel.init_from_env(os.environ.get("NERF_DIST_BACKEND"))   # default: nccl (= RCCL) on GPUs
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr, flush=True)
        sys.exit(2)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # what the communicator is, from inside it (a collective: every rank): backend, ranks that answered an all-reduce, RCCL
    # version, each rank's device.  A job whose communicator does not hold --gpus ranks never reaches the timed region.
    args.comm = parallel.comm_info(dev)
    if args.comm["world_size_seen"] != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the communicator's all-reduce saw {args.comm['world_size_seen']} ranks", file=sys.stderr, flush=True)
        sys.exit(2)
    _native.check(_native.lib().nerf_set_option(b"mlp_variant", args.mlp_variant))

    H = W = args.hw
    imgs, poses, rposes, hwf, K = synthetic.make_dataset(H, W, args.train_images, seed=0, device=dev)
    if args.config == "ngp":
        return bench_ngp(args, imgs, poses, rposes, K, rank, world, dev)
    # render chunk of this rank: a contiguous slice of a render pose's pixel list, resident on the device
    lo, _ = parallel.shard_range(H * W, rank, world)
    lo = min(lo, H * W - args.render_rays)
    ridx = torch.arange(lo, lo + args.render_rays, device=dev, dtype=torch.int64)
    rrays = ray.gen_rays(H, W, K, rposes[40][:3, :4], 2.0, 6.0, ridx)

    from nerf_meets_mlx_amd import sampling
    from nerf_meets_mlx_amd.rendering import render

    NI = args.n_importance
    n_fine = 64 + NI
    spr = (64 + n_fine) if NI > 0 else 64

    def barrier():
        torch.cuda.synchronize()
        parallel.barrier()
        torch.cuda.synchronize()

    def measure(precision, steps, warmup, n_rand=None):
        """`steps` timed steps (train N_rand rays + render one chunk) of a fresh Trainer at `precision`, after `warmup`
        untimed ones, bracketed by barrier + synchronize; the dominant kernel (fused MLP forward of the render fine
        pass) and both phases are timed with events on the launch stream inside the timed region."""
        # seed 4: both networks start with sigma > 0 (a net whose raw sigma starts negative everywhere has an exactly
        # zero gradient under the reference's formulas and never trains -- DESIGN.md section 7)
        n_rand = args.n_rand if n_rand is None else n_rand
        tr = Trainer(imgs, poses, K, N_rand=n_rand, n_depth_samples=64, N_importance=NI, seed=4, device=dev,
                     chunk=args.render_rays, precision=precision)
        ev, phase = [], {"train": [], "render": []}
        parallel.comm_timing = [] if world > 1 else None        # (start, end) events around every gradient all-reduce

        def timed_fine_query(r, zf):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            raw = (tr.fine or tr.coarse).query(r, zf, ref_quirks=True)
            e1.record()
            ev.append((e0, e1))
            return raw

        def render_chunk():
            z = sampling.sample_coarse(rrays, 64)
            if NI == 0:
                raw = timed_fine_query(rrays, z) if tr.fine is None else tr.coarse.query(rrays, z)
                return render.composite(raw, z, rrays, 0.0, True, need_weights=False)[0]
            raw = tr.coarse.query(rrays, z)
            _, _, _, w, _ = render.composite(raw, z, rrays, 0.0, True)
            u = torch.rand(rrays.shape[0], NI, device=dev, generator=tr.gen)
            _, zf = sampling.importance_sample(z, w, NI, u=u)
            raw = timed_fine_query(rrays, zf)
            return render.composite(raw, zf, rrays, 0.0, True, need_weights=False)[0]

        def step():
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
            out = tr.train_step()
            e[1].record()
            rgb = render_chunk()
            e[2].record()
            phase["train"].append((e[0], e[1])); phase["render"].append((e[1], e[2]))
            return out, rgb

        for _ in range(warmup):
            step()
        ev.clear(); phase["train"].clear(); phase["render"].clear()
        if parallel.comm_timing is not None:
            parallel.comm_timing.clear()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            out, rgb = step()
        barrier()
        dt = time.perf_counter() - t0
        comm_ms, rank_ms = None, None
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            # every rank's own wall time of the timed region, next to the MAX the contract asks for: the first SCALE record can
            # then separate rank skew (min vs max) from collective time (comm_ms_per_step)
            every = [torch.zeros_like(t) for _ in range(world)]
            torch.distributed.all_gather(every, t)
            rank_ms = [float(x[0]) / steps * 1e3 for x in every]
            cm = parallel.comm_timing
            parallel.comm_timing = None                          # the two reductions below are not part of the step
            # collective time per step on this rank (events on the stream the collectives are issued on; includes waiting
            # for the slowest rank) -> MAX over ranks like the step time
            c = torch.tensor([sum(a.elapsed_time(b) for a, b in cm) / steps], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            torch.distributed.all_reduce(c, op=torch.distributed.ReduceOp.MAX)
            dt, comm_ms = float(t[0]), float(c[0])
        assert torch.isfinite(rgb).all() and torch.isfinite(out["loss_coarse"]).all()
        t_train = float(np.mean([a.elapsed_time(b) for a, b in phase["train"]])) * 1e-3
        t_render = float(np.mean([a.elapsed_time(b) for a, b in phase["render"]])) * 1e-3
        return {"dt": dt, "steps": steps, "value": (n_rand + args.render_rays) * world * steps / dt, "n_rand": n_rand,
                "ms_per_step": dt / steps * 1e3, "k_ms": float(np.mean([a.elapsed_time(b) for a, b in ev])),
                "t_train": t_train, "t_render": t_render, "loss_coarse": float(out["loss_coarse"]),
                "loss_fine": float(out.get("loss_fine", torch.zeros(1))), "comm_ms_per_step": comm_ms, "rank_ms_per_step": rank_ms,
                "trainer": tr}

    def frame_leg(tr):
        """ONE full frame through the PUBLIC API, `render.render(H, W, K, chunk=32768, c2w=...)` (rendering/render.py:268-345 of
        the reference): rays generated inside, 20 chunks of 32768 (the last one ragged at 800 x 800), coarse + importance +
        fine per chunk, every output of the reference's return list (rgb / disp / acc + extras z_vals, weights, *_coarse)
        materialised and reshaped to the ray grid.  One untimed frame first (allocator warm-up), then one timed."""
        from nerf_meets_mlx_amd.models.NeRF import NetworkQuery
        kw = {"network_query_fn": NetworkQuery(None, None, 1024 * 64, True), "render_rays_func": render.render_rays_eval,
              "network_coarse": tr.coarse, "network_fine": tr.fine, "n_depth_samples": 64, "N_importance": NI,
              "white_bkgd": True, "perturb": False, "raw_noise_std": 0, "lindisp": False}
        c2w = rposes[40][:3, :4]
        for timed in (False, True):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rgb, disp, acc, extras = render.render(H, W, K, chunk=1024 * 32, c2w=c2w, ndc=False, near=2.0, far=6.0,
                                                   use_viewdirs=True, **kw)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        assert tuple(rgb.shape) == (H, W, 3) and torch.isfinite(rgb).all() and "z_vals" in extras and "weights" in extras
        return {"api": "render.render(H, W, K, chunk=32768, c2w=..., **render_kwargs_test)", "rays": H * W, "seconds": dt,
                "rays_per_s": H * W / dt, "chunks": (H * W + 32767) // 32768,
                "outputs": ["rgb_map", "disp_map", "acc_map"] + sorted(extras.keys())}

    HP = args.precision                                            # precision of the headline step (default 22)
    m = measure(HP, args.steps, args.warmup)                       # THE timed region of the contract
    dt, value = m["dt"], m["value"]
    # dominant kernel: fused MLP forward over render_rays x 192 samples
    flop = FLOP_PER_SAMPLE_FWD * args.render_rays * n_fine

    def derived(mm, prec, n_rand):
        """per-phase throughput (events on the launch stream) and algorithmic MFMA fractions (train = 3x forward
        FLOPs over 64 + n_fine samples, render = 1x; the repeated coarse forward of __test_nerf.py:270 is not credited)
        against the matrix peak of the precision: 16 -> dense bf16; 32 -> fp32 MFMA; 22 -> THREE 16-bit MFMAs per float32
        product, i.e. a third of the dense 16-bit peak in algorithmic FLOPs (executed matrix FLOPs = 3 x algorithmic)."""
        pk = PEAK_TFLOPS[prec]
        a = flop / (mm["k_ms"] * 1e-3) / 1e12
        d = {"value": mm["value"], "unit": "rays/s", "steps": mm["steps"], "ms_per_step": mm["ms_per_step"],
             "train_rays_per_s": n_rand / mm["t_train"], "render_rays_per_s": args.render_rays / mm["t_render"],
             "train_mfma_frac": 3 * FLOP_PER_SAMPLE_FWD * spr * n_rand / mm["t_train"] / 1e12 / pk,
             "render_mfma_frac": FLOP_PER_SAMPLE_FWD * spr * args.render_rays / mm["t_render"] / 1e12 / pk,
             "train_hbm_frac": TRAIN_BYTES[prec] * spr * n_rand / mm["t_train"] / 8e12,
             "loss_coarse": mm["loss_coarse"], "loss_fine": mm["loss_fine"], "dtype": DTYPE[prec],
             "roofline": {"bound": "mfma", "kernel": KERNEL[prec] + " (render fine pass)", "achieved": a, "peak": pk,
                          "unit": "TFLOP/s", "frac": a / pk, "ms_per_launch": mm["k_ms"],
                          "samples_per_launch": args.render_rays * n_fine}}
        if prec == 22:
            d["roofline"]["executed_mfma_tflops"] = 3 * a
            d["roofline"]["peak_note"] = ("algorithmic float32 FLOPs against 2500 / 3 TFLOP/s: every float32 product is three "
                                          "v_mfma_f32_16x16x32_f16 (hi hi + hi lo + lo hi) of the 2500 TFLOP/s dense 16-bit pipe")
        return d

    hd = derived(m, HP, args.n_rand)
    # HBM bytes per launch of that kernel: NOT measured by this run -- taken from the newest committed PMC passes of this
    # same command (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs, tools/collect_profiles.sh) when their kernel and
    # samples-per-launch match this workload; labelled as cached in the line.  null otherwise.
    traffic, traffic_src = None, None
    tpath = _latest_profile("r0*_pmc_traffic.json")
    if tpath:
        try:
            with open(tpath) as fp:
                pm = json.load(fp)
            if pm.get("samples_per_launch") == args.render_rays * n_fine and KERNEL[HP].split("<")[0] in pm.get("kernel", ""):
                traffic = pm["hbm_bytes_per_launch"]
                traffic_src = f"profiles/{os.path.basename(tpath)} (cached PMC run of this command, not this run; " + pm.get("note", "FETCH_SIZE + WRITE_SIZE") + ")"
        except (OSError, ValueError, KeyError):
            pass
    # matrix-pipe busy cycles / all SIMD cycles of that kernel (same caveat: cached PMC pass), null if absent
    busy, busy_src = None, None
    bpath = _latest_profile("r0*_pmc_mfma_lds.csv")
    if bpath and args.mlp_variant in (0, 4) and NI == 128 and args.render_rays == 32768:
        busy = _read_pmc_busy(bpath, KERNEL[HP])
        busy_src = f"profiles/{os.path.basename(bpath)} (cached PMC run)" if busy is not None else None
    mode_text = {
        22: "HEADLINE `value` = the REFERENCE-TOLERANCE mode on the 16-bit matrix pipe (precision 22): every float32 GEMM operand "
            "as a (hi, lo) pair of 16-bit numbers, three MFMAs per float32 product, fp32 accumulate -- split-bf16 training "
            "(csrc/mlp_s16.hip), split-fp16 rendering (csrc/mlp22.hip); held to the SAME fixture tolerances as the float32 "
            "kernels (1e-4 of the output scale vs the reference's own outputs; measured 1e-5 training / 2e-6 rendering).  The "
            "same step with literal float32 operands on the fp32 MFMA is `reference_precision_value`, the declared "
            "reduced-precision bf16 mode is the `bf16` leg (N = 1 line)",
        16: "HEADLINE `value` = the DECLARED REDUCED-PRECISION MODE (bf16 MFMA operands, fp32 accumulate; the reference computes "
            "in float32)",
        32: "HEADLINE `value` = the reference's float32 arithmetic on the fp32 MFMA"}[HP]
    line = {
        "metric": f"train+render rays/sec on Lego {H}x{W} (synthetic), " + (f"coarse+fine 64+{NI}" if NI > 0 else "coarse-only 64"),
        "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPE[HP], "operand_significand_bits": OPERAND_BITS[HP], "data": "synthetic",
        "config": {"workload": (f"configs[2]: Lego {H}x{W} coarse+fine NeRF (64+{NI} importance samples), " if NI > 0 else
                                f"configs[1]: Lego {H}x{W} coarse-only NeRF (64 samples/ray), ")
                               + f"step = train N_rand={args.n_rand} rays + render chunk {args.render_rays} rays per GPU; " + mode_text,
                   "n_rand_per_gpu": args.n_rand, "render_rays_per_gpu": args.render_rays, "parallelism": f"rays x{world}",
                   "precision": HP},
        "comm_ms_per_step": m["comm_ms_per_step"],
        # the communicator as the ranks saw it before the timed region (parallel.comm_info) + what the data path sends through it:
        # one sum all-reduce of the flat float32 gradient buffer per network step (coarse, fine), nothing else
        "comm": dict(args.comm, allreduce_bytes_per_step=(0 if world == 1 else 595844 * 4 * (2 if NI > 0 else 1)),
                     collectives_per_step=(0 if world == 1 else (2 if NI > 0 else 1)),
                     overlap="collectives + Adam on a comm stream; the fine network's run under the next iteration's coarse pass (engine/trainer.py)"),
        # per-rank wall time per step of the timed region (ms_per_step above is their MAX); null at N = 1
        "rank_ms_per_step": (None if m["rank_ms_per_step"] is None else
                             {"min": min(m["rank_ms_per_step"]), "max": max(m["rank_ms_per_step"]), "ranks": m["rank_ms_per_step"]}),
        "train_rays_per_s_per_gpu": hd["train_rays_per_s"], "render_rays_per_s_per_gpu": hd["render_rays_per_s"],
        "train_mfma_frac": hd["train_mfma_frac"], "render_mfma_frac": hd["render_mfma_frac"],
        # the training phase against its other roofline: activations + dZ written once and read once by the weight-gradient
        # jobs + sign-bit words (bf16: 21.4 KB per sample; split bf16: hi + lo blocks, 42.7 KB)
        "train_hbm_frac": hd["train_hbm_frac"],
        "loss_coarse": m["loss_coarse"], "loss_fine": m["loss_fine"],
        "roofline": dict(hd["roofline"], traffic=traffic, traffic_unit="bytes/launch (PMC)", traffic_source=traffic_src,
                         mfma_busy_cycles_frac=busy, mfma_busy_source=busy_src,
                         algorithmic_bytes=args.render_rays * (n_fine * 20 + 44)),
    }
    if world == 1 and not args.no_extra_legs:
        # ---- sustained: the same headline step for >= args.sustain_seconds (the burst above is ~2 s of a power-limited
        # kernel; this is what the chip holds, and long enough for an SMI sampler to see the GPU busy)
        n_sus = max(args.steps, int(np.ceil(args.sustain_seconds * 1.05 / (dt / args.steps))))
        ms = measure(HP, n_sus, 2)
        if ms["dt"] < args.sustain_seconds:                      # the burst's step time underestimated it (first-run effects): once more, scaled
            n_sus = int(np.ceil(n_sus * args.sustain_seconds * 1.1 / ms["dt"]))
            ms = measure(HP, n_sus, 2)
        ms.pop("trainer", None)
        line["sustained"] = {"seconds": ms["dt"], "steps": n_sus, "value": ms["value"], "unit": "rays/s",
                             "ms_per_step": ms["ms_per_step"], "ms_per_launch": ms["k_ms"],
                             "roofline_frac": flop / (ms["k_ms"] * 1e-3) / 1e12 / PEAK_TFLOPS[HP],
                             "train_rays_per_s": args.n_rand / ms["t_train"], "render_rays_per_s": args.render_rays / ms["t_render"],
                             "ratio_to_burst": ms["value"] / value, "dtype": DTYPE[HP]}
        # ---- lego.txt's batch: the reference's configs/lego.txt trains with N_rand = 1024 (the headline uses the argparse
        # default 4096, config_parser.py:17); same step, launch overheads weigh more
        if args.n_rand != 1024:
            ml = measure(HP, args.steps, args.warmup, n_rand=1024)
            ml.pop("trainer", None)
            dl = derived(ml, HP, 1024)
            line["lego_batch"] = {"n_rand_per_gpu": 1024, **{k: dl[k] for k in ("value", "unit", "steps", "ms_per_step", "train_rays_per_s",
                                                                                   "render_rays_per_s", "train_mfma_frac", "dtype")}}
        # ---- the other two precisions of the same step: literal float32 operands on v_mfma_f32_32x32x2_f32 (the REFERENCE's
        # arithmetic, models/NeRF.py:201-243 runs in MLX float32) and the declared reduced-precision bf16 mode
        trainers = {HP: m["trainer"]}
        for prec, key, steps in ((32, "fp32", args.fp32_steps), (16, "bf16", args.steps), (22, "f32tol", args.steps)):
            if prec == HP:
                continue
            mo = measure(prec, steps, 2 if prec == 32 else args.warmup)
            trainers[prec] = mo.pop("trainer")
            line[key] = derived(mo, prec, args.n_rand)
        r32 = line["fp32"] if HP != 32 else hd
        # the literal-float32 measurement where the driver parses it: top level + inside `roofline`
        line["reference_precision_value"] = r32["value"]
        line["reference_precision_unit"] = "rays/s (same step, float32 operands on the fp32 MFMA: models/NeRF.py:201-243 of the reference runs in MLX float32)"
        line["roofline"]["reference_precision"] = dict(r32["roofline"], dtype="f32", value=r32["value"], value_unit="rays/s", steps=r32["steps"],
                                                       ms_per_step=r32["ms_per_step"], achieved_unit="TFLOP/s",
                                                       train_rays_per_s=r32["train_rays_per_s"], render_rays_per_s=r32["render_rays_per_s"])
        # ---- frame: one full 800 x 800 frame through the public render.render() API, all three precisions
        line["frame"] = {DTYPE_SHORT[p_]: frame_leg(trainers[p_]) for p_ in (22, 16, 32) if p_ in trainers}
        trainers.clear()
        # ---- ngp: BASELINE configs[4] (hash grid + 2x64 MLP), its own step and roofline (the table-gradient scatter)
        # value = the reference-tolerance mode (precision 22); the declared reduced-precision bf16 mode rides along as `bf16`
        ngp_line = measure_ngp(args, imgs, poses, rposes, K, rank, world, dev, extra_bf16=True)
        line["ngp"] = {k: ngp_line[k] for k in ("value", "unit", "ms_per_step", "steps", "train_rays_per_s_per_gpu",
                                                "render_rays_per_s_per_gpu", "roofline", "dtype", "loss_coarse", "bf16") if k in ngp_line}
        line["ngp"]["workload"] = ngp_line["config"]["workload"]
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: re-run this command line under torch.distributed.run with N
    ranks (child processes; the parent never initialises the GPU, and nothing is exec'ed after GPU init)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")         # dmabuf IPC only on this pool (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def _read_pmc_busy(path, kernel_substr):
    """mfma_busy_frac_of_cycles of the first row whose kernel name contains kernel_substr.  Kernel names contain commas
    (template arguments); rows are parsed from the right so that both quoted and unquoted files read correctly."""
    try:
        with open(path) as fp:
            lines = [ln.rstrip("\n") for ln in fp if ln.strip()]
    except OSError:
        return None
    import csv
    header = next(csv.reader([lines[0]]))
    k = len(header)
    col = header.index("mfma_busy_frac_of_cycles") if "mfma_busy_frac_of_cycles" in header else None
    if col is None:
        return None
    for ln in lines[1:]:
        row = next(csv.reader([ln]))
        if len(row) != k:                                     # unquoted name with commas: split the numeric tail off
            tail = ln.rsplit(",", k - 1)
            row = [tail[0].strip('"')] + tail[1:]
        if kernel_substr.replace(" ", "") in row[0].replace(" ", ""):
            try:
                return float(row[col])
            except ValueError:
                return None
    return None


def _latest_profile(pattern):
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return c[-1] if c else None


def bench_ngp(args, imgs, poses, rposes, K, rank, world, dev):
    line = measure_ngp(args, imgs, poses, rposes, K, rank, world, dev, extra_bf16=(world == 1 and not args.no_extra_legs))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline_ngp(args)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def measure_ngp(args, imgs, poses, rposes, K, rank, world, dev, precision=None, extra_bf16=False):
    """BASELINE configs[4]: hash grid (16 levels x 2^19 x 2) + SH + NeRF 2x64, 64 samples per ray, coarse-only loop.
    Same step structure as the headline bench (train N_rand rays + render one chunk); the dominant kernel is the
    hash-grid gradient scatter (float atomics), reported against the HBM roofline with its algorithmic bytes.
    precision: of the 2 x 64 network and its inputs (default: --precision when it is 16 or 22, else 22 -- the reference
    tolerance: float32 gathers + interpolation, split-bf16 MLP); extra_bf16: also measure the declared bf16 mode as `bf16`."""
    from nerf_meets_mlx_amd import parallel
    from nerf_meets_mlx_amd.engine.ngp import NGPTrainer
    from nerf_meets_mlx_amd.rendering import ray
    H = W = args.hw
    if precision is None:
        precision = args.precision if args.precision in (16, 22) else 22
    tr = NGPTrainer(imgs, poses, K, N_rand=args.n_rand, n_depth_samples=64, seed=4, device=dev, chunk=args.render_rays,
                    precision=precision)
    tr.field.timing = []
    lo, _ = parallel.shard_range(H * W, rank, world)
    lo = min(lo, H * W - args.render_rays)
    ridx = torch.arange(lo, lo + args.render_rays, device=dev, dtype=torch.int64)
    rrays = ray.gen_rays(H, W, K, rposes[40][:3, :4], 2.0, 6.0, ridx)
    phase = {"train": [], "render": []}

    def step():
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record(); out = tr.train_step()
        tr._join_comm()          # N > 1: the table all-gathers queued on the comm stream belong to the TRAINING phase (stream-side wait)
        e[1].record(); rgb = tr.render_rays(rrays); e[2].record()
        phase["train"].append((e[0], e[1])); phase["render"].append((e[1], e[2]))
        return out, rgb

    def barrier():
        torch.cuda.synchronize(); parallel.barrier(); torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    tr.field.timing.clear(); phase["train"].clear(); phase["render"].clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, rgb = step()
    barrier()
    dt = time.perf_counter() - t0
    rank_ms = None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        torch.distributed.all_gather(every, t)
        rank_ms = [float(x[0]) / args.steps * 1e3 for x in every]
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t[0])
    assert torch.isfinite(rgb).all() and torch.isfinite(out["loss_coarse"]).all()
    t_train = float(np.mean([a.elapsed_time(b) for a, b in phase["train"]])) * 1e-3
    t_render = float(np.mean([a.elapsed_time(b) for a, b in phase["render"]])) * 1e-3
    k_ms = float(np.mean([a.elapsed_time(b) for a, b in tr.field.timing]))
    M = args.n_rand * 64
    alg = M * (12 + 128 + 16 * 8 * 2 * 4 * 2)          # points + feature gradients + read-modify-write of 256 table floats
    line = {
        "metric": f"train+render rays/sec on Lego {H}x{W} (synthetic), hash grid 16x2^19x2 + SH3 + 2x64 MLP, 64 samples/ray",
        "value": (args.n_rand + args.render_rays) * world * args.steps / dt, "unit": "rays/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": NGP_DTYPE[precision],
        "operand_significand_bits": {"render": 16 if precision == 22 else 8, "train": 16 if precision == 22 else 8}, "data": "synthetic",
        "config": {"workload": f"configs[4]: Lego {H}x{W} hash-grid + tiny fused MLP, step = train N_rand={args.n_rand} rays + render "
                               f"chunk {args.render_rays} rays per GPU, 64 samples per ray; precision {precision}"
                               + (" (the reference's float32 tolerance)" if precision == 22 else " (declared reduced precision)"),
                   "n_rand_per_gpu": args.n_rand, "render_rays_per_gpu": args.render_rays, "parallelism": f"rays x{world}",
                   "precision": precision},
        "rank_ms_per_step": None if rank_ms is None else {"min": min(rank_ms), "max": max(rank_ms), "ranks": rank_ms},
        # the communicator as the ranks saw it (parallel.comm_info) + the bytes of the table / MLP gradient exchange per step
        "comm": dict(getattr(args, "comm", None) or parallel.comm_info(dev),
                     allreduce_bytes_per_step=(0 if world == 1 else int(tr.field.enc.grad.numel() * tr.field.enc.grad.element_size()
                                                                        + tr.field.mlp.n_params * 4)),
                     table_sync=tr.table_sync if world > 1 else None),
        "train_rays_per_s_per_gpu": args.n_rand / t_train, "render_rays_per_s_per_gpu": args.render_rays / t_render,
        "loss_coarse": float(out["loss_coarse"]),
        "roofline": {"bound": "hbm", "kernel": "hashgrid_bwd_combine_kernel<2> (levels 0-4) + hashgrid_bwd_kernel<2, 4> (table gradient scatter, "
                               + ("int64 fixed-point integer atomics: deterministic" if tr.field.deterministic else "float atomics") + ")",
                     "achieved": alg / (k_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": alg / (k_ms * 1e-3) / 1e9 / 8000.0, "traffic": None, "algorithmic_bytes": alg,
                     "ms_per_launch": k_ms, "samples_per_launch": M,
                     "note": "atomic-request-rate bound (device-scope atomics execute memory-side, ~70 G/s on the collision-free fine levels; "
                             "8-byte integer atomics cost what 4-byte float atomics cost); algorithmic bytes count a read-modify-write of 256 float32 "
                             "table entries per sample"},
    }
    if extra_bf16 and precision != 16:
        del tr
        torch.cuda.empty_cache()
        b = measure_ngp(args, imgs, poses, rposes, K, rank, world, dev, precision=16)
        line["bf16"] = {k: b[k] for k in ("value", "unit", "ms_per_step", "steps", "train_rays_per_s_per_gpu", "render_rays_per_s_per_gpu",
                                          "dtype", "loss_coarse")}
        line["bf16"]["scatter_ms_per_launch"] = b["roofline"]["ms_per_launch"]
    return line


def _host_cores():
    """Host share of a 1-GPU box: the scheduler affinity when it is set, never more than 16 threads (an 8-GPU host
    exposes 256 logical CPUs to every process; oversubscribing them makes torch-CPU crawl)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return max(1, min(cores, 16))


def cpu_baseline(args):
    """The CPU oracle (op-for-op torch-CPU fp32 restatement of the reference path, `kind: port`) at the size SURVEY
    8(d) specifies: training at B = 1024 rays (lego.txt N_rand), 3 warm-up + 20 timed steps (about 100 s on 16 cores; only
    a safety cap, --cpu-train-cap, can cut it short), and a render slice of 65 536 rays in 4096-ray chunks (coarse 64 + fine
    64+128 like the GPU step; coarse-only when --n-importance 0), time-boxed by --cpu-seconds: the steps / rays actually
    timed are in `sample`.  `value` combines the two legs in the GPU step's train : render ray mix."""
    from oracle import nerf_oracle as O
    cores = _host_cores()
    torch.set_num_threads(cores)
    NI = args.n_importance
    arch = O.NerfArch()
    tr = O.OracleTrainer(arch, 64, NI, seed=4)
    b_train, n_render, chunk = 1024, 65536, 4096
    g = torch.Generator().manual_seed(0)
    o = torch.nn.functional.normalize(torch.randn(n_render, 3, generator=g), dim=-1) * 4.0
    d = -o / 4.0 + 0.2 * torch.randn(n_render, 3, generator=g)
    y = torch.rand(b_train, 3, generator=g)
    un = max(NI, 1)
    for _ in range(3):                                                                   # 3 warm-up steps
        tr.step(o[:b_train], d[:b_train], y, torch.rand(b_train, un, generator=g))
    t0 = time.perf_counter()
    steps = 0
    while steps < 20 and time.perf_counter() - t0 < args.cpu_train_cap:    # 20 timed iterations after 3 warm-ups
        tr.step(o[:b_train], d[:b_train], y, torch.rand(b_train, un, generator=g))
        steps += 1
    t_train = time.perf_counter() - t0
    pc = O.unflatten_params(arch, tr.pc.detach())
    pf = O.unflatten_params(arch, tr.pf.detach()) if tr.pf is not None else None
    t1 = time.perf_counter()
    done = 0
    with torch.no_grad():
        while done < n_render and (done < 2 * chunk or time.perf_counter() - t1 < args.cpu_seconds):
            rays = O.pack_rays(o[done:done + chunk], d[done:done + chunk], 2.0, 6.0)
            if NI > 0:
                O.render_rays_eval(arch, pc, pf, rays, 64, NI, torch.rand(rays.shape[0], NI, generator=g), True)
            else:
                O.render_rays(arch, pc, rays, 64, True)
            done += rays.shape[0]
    t_render = time.perf_counter() - t1
    train_rps, render_rps = b_train * steps / t_train, done / t_render
    mix_t = args.n_rand / train_rps + args.render_rays / render_rps               # CPU seconds for one GPU-sized step
    return {"value": (args.n_rand + args.render_rays) / mix_t, "unit": "rays/s", "cores": cores, "kind": "port",
            "train_rays_per_s": train_rps, "render_rays_per_s": render_rps,
            "sample": f"train: {steps} steps x {b_train} rays ({t_train:.1f} s); render: {done} rays in {chunk}-ray chunks "
                      f"({t_render:.1f} s); coarse 64" + (f" + fine 64+{NI}" if NI > 0 else "") + ", torch-CPU fp32 oracle; "
                      f"value = rays of one GPU step (train {args.n_rand} + render {args.render_rays}) / CPU time for them",
            "seconds": t_train + t_render}


def cpu_baseline_ngp(args):
    """configs[4] on the host cores: OracleNGP (hash grid 16 x 2^19 x 2 + SH3 + 2 x 64 MLP, autograd) training at
    B = 1024 rays and a 65 536-ray render slice, time-boxed like cpu_baseline."""
    from oracle import nerf_oracle as O
    cores = _host_cores()
    torch.set_num_threads(cores)
    res = O.hashgrid_resolutions(16, 16, 2048)
    g = torch.Generator().manual_seed(0)
    tables = (torch.rand(16, 1 << 19, 2, generator=g) * 2 - 1) * 1e-4
    ng = O.OracleNGP(tables, res, seed=4, n_samples=64)
    b_train, n_render, chunk = 1024, 65536, 8192
    o = torch.nn.functional.normalize(torch.randn(n_render, 3, generator=g), dim=-1) * 4.0
    d = -o / 4.0 + 0.2 * torch.randn(n_render, 3, generator=g)
    y = torch.rand(b_train, 3, generator=g)
    ng.step(o[:b_train], d[:b_train], y)
    t0 = time.perf_counter()
    steps = 0
    while steps < 20 and (steps < 3 or time.perf_counter() - t0 < args.cpu_seconds):
        ng.step(o[:b_train], d[:b_train], y)
        steps += 1
    t_train = time.perf_counter() - t0
    t1 = time.perf_counter()
    done = 0
    with torch.no_grad():
        while done < n_render and (done < 2 * chunk or time.perf_counter() - t1 < args.cpu_seconds):
            ng.render(O.pack_rays(o[done:done + chunk], d[done:done + chunk], 2.0, 6.0))
            done += chunk
    t_render = time.perf_counter() - t1
    train_rps, render_rps = b_train * steps / t_train, done / t_render
    mix_t = args.n_rand / train_rps + args.render_rays / render_rps
    return {"value": (args.n_rand + args.render_rays) / mix_t, "unit": "rays/s", "cores": cores, "kind": "port",
            "train_rays_per_s": train_rps, "render_rays_per_s": render_rps,
            "sample": f"train: {steps} steps x {b_train} rays ({t_train:.1f} s); render: {done} rays ({t_render:.1f} s); "
                      "hash grid 16x2^19x2 + SH3 + 2x64, 64 samples/ray, torch-CPU fp32 oracle (OracleNGP)",
            "seconds": t_train + t_render}


if __name__ == "__main__":
    main()
