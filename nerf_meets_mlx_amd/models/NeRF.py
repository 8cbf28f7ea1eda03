"""The 8x256 NeRF MLP, model factory and optimiser with the reference's call surface
(`mlx_nerf/models/NeRF.py:10-243`), backed by the fused gfx950 kernels of csrc/mlp.hip.

Parameters live in ONE flat float32 device buffer per network (layout: include/nerf_hip.h);
the kernels stream a bf16 MFMA-fragment image of it (`packed`) that is rebuilt after every
optimiser step.  `forward(x)` is the reference's `NeRF.forward` on embedded rows; the render
path uses `query(rays, z)` which fuses pts = o + z d, both positional encodings and all 12
layers into one launch.
"""
import ctypes as C
import math
from typing import Dict, List, Optional

import numpy as np
import torch

from .. import _native as N
from . import embedding

_LAYERS = None


def layer_shapes(n_layers=8, width=256, cin=63, cdir=27, skips=(4,), use_viewdirs=True, cout=4):
    """(name, out, in) in flat-buffer order (include/nerf_hip.h "Parameter layout")."""
    s = [("pos0", width, cin)]
    for i in range(n_layers - 1):
        s.append((f"pos{i + 1}", width, width + cin if i in skips else width))
    if use_viewdirs:
        s += [("feature", width, width), ("alpha", 1, width), ("dir0", width // 2, width + cdir), ("rgb", 3, width // 2)]
    else:
        s += [("output", cout, width)]
    return s


class NeRF:
    """Same constructor arguments as `mlx_nerf/models/NeRF.py:160-199`.  Initialisation is
    mlx.nn.Linear's: weight and bias ~ U(-1/sqrt(in), 1/sqrt(in)) (seeded numpy stream).
    precision (ours; the reference computes in MLX float32):
      22 (DEFAULT) = the reference's float32 TOLERANCE on the 16-bit matrix pipe: every float32 GEMM operand as a (hi, lo) pair
           of 16-bit numbers, three MFMAs per product, fp32 accumulate.  8 x 256 view model: split-fp16 inference
           (csrc/mlp22.hip, ~3.3 x the fp32 MFMA's speed) and split-bf16 training (csrc/mlp_s16.hip, forward 1e-5 / gradients
           3e-5 of the fp32 oracle at ~2.5 x); image-fitting and 2 x 64 models: split bf16 for both (csrc/mlp_s16x.hip).
           Every fixture of the reference's own float32 outputs is met at the fp32 bar (1e-4) in this mode;
      32 = float32 operands on the fp32 MFMA (csrc/mlp32.hip; 8 x 256 view and image models);
      16 = bf16 MFMA operands with fp32 accumulate: a declared REDUCED-precision mode (opt-in; ~1e-2 of the output scale).
    It is part of the model (`nerf_mlp_arch.precision`): weight image, workspaces and every launch of this object use it;
    models of different precision can be used side by side."""

    def __init__(self, n_layers=8, width_layers=256, channel_input=3, channel_input_views=3, channel_output=4,
                 list_skip_connection_layers=[4], is_use_view_directions=False, device="cuda", seed: Optional[int] = None,
                 precision: int = 22):
        self.D, self.W = n_layers, width_layers
        self.channel_input_pos, self.channel_input_dir = channel_input, channel_input_views
        self.list_skip_connection_layers = list(list_skip_connection_layers)
        self.is_use_view_directions = is_use_view_directions
        self.device = torch.device(device)
        self.shapes = layer_shapes(n_layers, width_layers, channel_input, channel_input_views,
                                   tuple(self.list_skip_connection_layers), is_use_view_directions, channel_output)
        self.n_params = sum(o * i + o for _, o, i in self.shapes)
        self.arch = N.MlpArch(n_layers, width_layers, channel_input, channel_input_views,
                              self.list_skip_connection_layers[0] if len(self.list_skip_connection_layers) == 1 else -1,
                              int(bool(is_use_view_directions)), channel_output, int(precision))
        if precision not in (16, 22, 32):
            raise ValueError("NeRF: precision must be 16 (bf16 MFMA operands), 32 (fp32 MFMA) or 22 (float32 tolerance on the "
                             "16-bit matrix pipe: split-bf16 training, split-fp16 inference)")
        self.precision = int(precision)
        self.out_dim = 4 if is_use_view_directions else channel_output
        rng = np.random.default_rng(seed)
        chunks = []
        for _, o, i in self.shapes:
            k = 1.0 / math.sqrt(i)
            chunks.append(rng.uniform(-k, k, size=(o, i)).astype(np.float32).reshape(-1))
            chunks.append(rng.uniform(-k, k, size=(o,)).astype(np.float32))
        self.params = torch.from_numpy(np.concatenate(chunks)).to(self.device)
        self.grads = torch.zeros_like(self.params)
        self._packed = None
        self._dirty = True
        self._packed_version = -1           # params._version the packed image was built from
        self._ws: Dict[str, torch.Tensor] = {}
        self.name: Optional[str] = None     # stable key for optimiser state / checkpoints ("coarse", "fine", ...)
        self._gen = 0                       # generation of the activation workspace (one per train-mode forward)
        self._acts_gen = -1
        self._acts_M = -1

    # ---- parameter views (the reference's tree: list_linears_pos / list_linears_dir / ... ) ----
    def parameters(self) -> Dict[str, object]:
        views, off = {}, 0
        for name, o, i in self.shapes:
            views[name] = {"weight": self.params[off:off + o * i].view(o, i), "bias": self.params[off + o * i:off + o * i + o]}
            off += o * i + o
        tree = {"list_linears_pos": [views[f"pos{l}"] for l in range(self.D)]}
        if self.is_use_view_directions:
            tree.update({"list_linears_dir": [views["dir0"]], "feature_linear": views["feature"],
                         "alpha_linear": views["alpha"], "rgb_linear": views["rgb"]})
        else:
            tree["output_linear"] = views["output"]
        return tree

    def trainable(self) -> torch.Tensor:
        """The flat parameter buffer as an autograd leaf (for nerf_meets_mlx_amd.autograd); same storage as `params`."""
        if not self.params.requires_grad:
            self.params.requires_grad_(True)
        return self.params

    def load_flat(self, flat: torch.Tensor):
        assert flat.numel() == self.n_params
        with torch.no_grad():
            self.params.copy_(flat.to(self.device, torch.float32).reshape(-1))
        self._dirty = True

    def mark_updated(self):
        self._dirty = True

    def activation_generation(self) -> int:
        return self._acts_gen

    # ---- packed bf16 fragment image ------------------------------------------------------------
    def packed(self) -> torch.Tensor:
        lib = N.lib()
        if self._packed is None:
            nbytes = lib.nerf_mlp_packed_bytes(C.byref(self.arch))
            if nbytes < 0:
                raise ValueError("libnerf_hip error -3: this NeRF architecture / precision has no HIP kernel (supported: "
                                 "n_layers=8, width=256, skips=[4] with in=63+27 + view head [precision 16, 22, 32], or in=40 without "
                                 "view head and out<=4 [16, 22, 32]; n_layers=2, width=64, skips=[] with in=32+16 + view head [16, 22])")
            self._packed = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        # re-pack when the master parameters changed: kernels that write through raw pointers (Adam) say so with
        # mark_updated(); torch in-place ops on `params` or on the views of parameters() bump the version counter
        if self._dirty or self.params._version != self._packed_version:
            N.check(lib.nerf_mlp_pack(C.byref(self.arch), N.ptr(self.params), N.ptr(self._packed), N.stream()))
            self._dirty = False
            self._packed_version = self.params._version
        return self._packed

    def _begin_train_pass(self, M: int) -> torch.Tensor:
        """The ONE activation workspace of this model, stamped with a new generation: a later train-mode forward
        overwrites it, and backward() refuses a generation that is no longer the stored one."""
        acts = self._workspace("acts", N.lib().nerf_mlp_acts_bytes(C.byref(self.arch), M))
        self._gen += 1
        self._acts_gen, self._acts_M = self._gen, M
        return acts

    def _workspace(self, key: str, nbytes: int) -> torch.Tensor:
        t = self._ws.get(key)
        if t is None or t.numel() < nbytes:
            t = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws[key] = t
        return t

    # ---- NeRF.forward(x) on embedded rows (models/NeRF.py:201-243) -----------------------------
    def forward(self, x: torch.Tensor, train: bool = False) -> torch.Tensor:
        """Embedded rows [M, C_in] -> [M, 4] (view model: rgb, alpha) or [M, channel_output].  train=True keeps the
        activations for `backward` (the reference gets gradients from nn.value_and_grad around this call)."""
        x = N.f32(x).reshape(-1, x.shape[-1])
        M = x.shape[0]
        out = torch.empty(M, self.out_dim, dtype=torch.float32, device=x.device)
        acts = None
        if train:
            acts = self._begin_train_pass(M)
        N.check(N.lib().nerf_mlp_forward_train(C.byref(self.arch), N.ptr(self.packed()), N.ptr(x), M, N.ptr(out),
                                               N.ptr(acts), N.stream()))
        return out

    __call__ = forward

    # ---- fused query: rays [B,11], z [B,n] -> raw [B,n,4] --------------------------------------
    def query(self, rays: torch.Tensor, z: torch.Tensor, ref_quirks: bool = True, train: bool = False) -> torch.Tensor:
        B, n = z.shape
        raw = torch.empty(B, n, 4, dtype=torch.float32, device=z.device)
        acts = None
        if train:
            acts = self._begin_train_pass(B * n)
        N.check(N.lib().nerf_query_fused(C.byref(self.arch), N.ptr(self.packed()), N.ptr(rays), N.ptr(z), B, n,
                                         0 if ref_quirks else 1, N.ptr(raw), N.ptr(acts), N.stream()))
        return raw

    def backward(self, d_raw: torch.Tensor, need_input_grad: bool = False, generation: Optional[int] = None):
        """Parameter gradients of the last `query / forward(..., train=True)`; overwrites self.grads.
        need_input_grad (2 x 64 model only): also returns dL/d(position features) [M, channel_input] for a trainable
        encoder in front of the network (the hash grid).
        generation: the value of `activation_generation()` right after the forward this gradient belongs to; when a
        later train-mode forward of the same model has replaced the stored activations this raises instead of returning
        gradients of the wrong graph."""
        d_raw = N.f32(d_raw)
        M = d_raw.numel() // self.out_dim
        if generation is not None and generation != self._acts_gen:
            raise RuntimeError(f"NeRF.backward: the activations of forward #{generation} were overwritten by train-mode "
                               f"forward #{self._acts_gen} of the same model (one activation workspace per model: run "
                               "backward before the next train-mode forward, or use a second model object)")
        assert M == self._acts_M, "backward() needs a matching query/forward(train=True) first"
        dz = self._workspace("dz", N.lib().nerf_mlp_dz_bytes(C.byref(self.arch), M))
        if need_input_grad:
            d_x = torch.empty(M, self.channel_input_pos, dtype=torch.float32, device=d_raw.device)
            N.check(N.lib().nerf_mlp_backward_inputs(C.byref(self.arch), N.ptr(self.packed()), N.ptr(self._ws["acts"]),
                                                     N.ptr(d_raw), M, N.ptr(dz), N.ptr(self.grads), N.ptr(d_x), N.stream()))
            return self.grads, d_x
        N.check(N.lib().nerf_mlp_backward(C.byref(self.arch), N.ptr(self.packed()), N.ptr(self._ws["acts"]),
                                          N.ptr(d_raw), M, N.ptr(dz), N.ptr(self.grads), N.stream()))
        return self.grads


def debug_layer(model: NeRF, kind: str, layer: int) -> torch.Tensor:
    """Test hook (`nerf_mlp_debug_read`): the stored activation ("acts") or dZ ("dz") of `layer` from the model's last
    train-mode forward / backward as a row-major [M, width] float32 tensor."""
    k = {"acts": 0, "dz": 1}[kind]
    w = N.lib().nerf_mlp_debug_width(C.byref(model.arch), k, layer)
    if w < 0:
        raise ValueError("debug_layer: no such (kind, layer) in this model's training stores (include/nerf_hip.h, nerf_mlp_debug_read)")
    M = model._acts_M
    out = torch.empty(M, w, dtype=torch.float32, device=model.device)
    N.check(N.lib().nerf_mlp_debug_read(C.byref(model.arch), N.ptr(model._ws[kind]), k, layer, M, N.ptr(out), N.stream()))
    return out


class Adam:
    """mlx.optimizers.Adam as the reference uses it (`models/NeRF.py:120`,
    `entrypoints/__test_nerf.py:134,144`): ONE optimiser object steps both networks; MLX keys
    its state by parameter-tree path, so coarse and fine (identical trees) share m and v
     -- `shared_state=True`.  No bias correction in mlx 0.7.0."""

    def __init__(self, learning_rate: float, betas=(0.9, 0.999), eps: float = 1e-8, bias_correction: bool = False,
                 shared_state: bool = True):
        self.learning_rate = learning_rate
        self.betas, self.eps = betas, eps
        self.bias_correction, self.shared_state = bias_correction, shared_state
        self.state: Dict[str, List[torch.Tensor]] = {}
        self.step_count: Dict[str, int] = {}
        self._auto_names: Dict[int, str] = {}

    def key_of(self, model) -> str:
        """State key: "shared" (Q7), else the model's `name`, else "model<k>" in order of first use -- never id(),
        so that a checkpoint written by one process can be read by another."""
        if self.shared_state:
            return "shared"
        name = getattr(model, "name", None)
        if name:
            return str(name)
        return self._auto_names.setdefault(id(model), f"model{len(self._auto_names)}")

    def update(self, model: NeRF, grads: Optional[torch.Tensor] = None, grad_scale: float = 1.0, zero_grads: bool = False):
        """zero_grads: clear the gradient buffer in the same pass (`nerf_adam_step_ex`: read g, write 0) -- for gradients
        that are ACCUMULATED by a scatter (the hash tables); an int64 `grads` tensor is taken as the fixed-point accumulators
        of the deterministic scatter (nerf_hashgrid_backward_rays_ex)."""
        g = model.grads if grads is None else grads
        key = self.key_of(model)
        if key not in self.state:
            self.state[key] = [torch.zeros_like(model.params), torch.zeros_like(model.params)]
        self.step_count[key] = self.step_count.get(key, 0) + 1
        m, v = self.state[key]
        half = getattr(model, "half", None)                  # fp16 shadow of the parameters (hash tables): written in the same pass
        if zero_grads or g.dtype == torch.int64 or half is not None:
            N.check(N.lib().nerf_adam_step_shadow(N.ptr(model.params), N.ptr(g), N.ptr(m), N.ptr(v), model.n_params,
                                                  float(self.learning_rate), float(self.betas[0]), float(self.betas[1]),
                                                  float(self.eps), int(self.bias_correction), self.step_count[key],
                                                  float(grad_scale), int(g.dtype == torch.int64), int(bool(zero_grads)),
                                                  N.ptr(half), N.stream()))
            if half is not None:
                model.mark_updated(shadow_written=True)
                return
        else:
            N.check(N.lib().nerf_adam_step(N.ptr(model.params), N.ptr(g), N.ptr(m), N.ptr(v), model.n_params,
                                           float(self.learning_rate), float(self.betas[0]), float(self.betas[1]),
                                           float(self.eps), int(self.bias_correction), self.step_count[key],
                                           float(grad_scale), N.stream()))
        model.mark_updated()

    def update_spans(self, model, spans, grad_scale: float = 1.0):
        """ONE optimiser step of `model` applied to the element ranges `spans` = [(lo, hi, grads[hi - lo]), ...] only (the
        shards of a flat buffer this rank owns: NGPTrainer's reduce-scatter update).  Moments of the other elements are left
        alone; the step count advances once.  float32 or int64 fixed-point gradients, not cleared."""
        key = self.key_of(model)
        if key not in self.state:
            self.state[key] = [torch.zeros_like(model.params), torch.zeros_like(model.params)]
        self.step_count[key] = self.step_count.get(key, 0) + 1
        m, v = self.state[key]
        half = getattr(model, "half", None)                  # fp16 shadow: the owned spans are written in the same pass; the caller
        for lo, hi, g in spans:                               # all-gathers the other ranks' spans and marks the shadow current
            assert g.numel() == hi - lo and g.is_contiguous()
            N.check(N.lib().nerf_adam_step_shadow(N.ptr(model.params[lo:hi]), N.ptr(g), N.ptr(m[lo:hi]), N.ptr(v[lo:hi]), hi - lo,
                                                  float(self.learning_rate), float(self.betas[0]), float(self.betas[1]),
                                                  float(self.eps), int(self.bias_correction), self.step_count[key],
                                                  float(grad_scale), int(g.dtype == torch.int64), 0,
                                                  N.ptr(half[lo:hi]) if half is not None else None, N.stream()))
        model.mark_updated()

    def state_dict(self) -> Dict[str, object]:
        return {"state": {k: [t.detach().cpu() for t in v] for k, v in self.state.items()},
                "step_count": dict(self.step_count), "learning_rate": float(self.learning_rate)}

    def load_state_dict(self, sd, device="cuda"):
        self.state = {k: [t.to(device, torch.float32).contiguous().clone() for t in v] for k, v in sd["state"].items()}
        self.step_count = {k: int(v) for k, v in sd.get("step_count", {}).items()}
        for k in self.state:
            self.step_count.setdefault(k, 0)
        if "learning_rate" in sd:
            self.learning_rate = float(sd["learning_rate"])


def inference_wrapper_batch(model, chunk):
    """`models/NeRF.py:10-22`."""
    if chunk is None:
        return model
    return lambda x: torch.cat([model.forward(x[i:i + chunk]) for i in range(0, x.shape[0], chunk)], dim=0)


def run_model(pos, embed_pos, dir, embed_dir, model, netchunk=64 * 1024):
    """Generic (unfused) query: embed all points, forward in `netchunk` slices
    (`models/NeRF.py:25-48`).  Mirrors the rank assertion at :31."""
    assert len(pos.shape) == 3, f"[ERROR] {pos.shape=} should have dimensions as: [n_rays, n_depth_samples, 3d position]!"
    B, n = pos.shape[0], pos.shape[1]
    x = embedding.embed(pos, embed_pos, dir, embed_dir)
    out = inference_wrapper_batch(model, netchunk)(x)
    return out.reshape(B, n, out.shape[-1])


class NetworkQuery:
    """`network_query_fn(inputs, viewdirs, model)` of `create_NeRF` (`models/NeRF.py:75-80`).
    Calling it takes the generic path above; `.fused(rays, z, model)` is the one-launch form the
    renderers in rendering/render.py use when they own the sampling."""

    def __init__(self, embed_pos, embed_dir, netchunk, ref_quirks=True):
        self.embed_pos, self.embed_dir, self.netchunk, self.ref_quirks = embed_pos, embed_dir, netchunk, ref_quirks

    def __call__(self, inputs, viewdirs, model):
        return run_model(inputs, self.embed_pos, viewdirs, self.embed_dir, model, netchunk=self.netchunk)

    def fused(self, rays, z, model, train=False):
        return model.query(rays, z, ref_quirks=self.ref_quirks, train=train)


def create_NeRF(args, device="cuda", ref_quirks: bool = True, seed: Optional[int] = 0, precision: int = 22):
    """Coarse (& fine) models, the query function, one Adam, and the render kwargs
    (`models/NeRF.py:51-158`).  precision: see `NeRF` -- the default (22) meets the reference's float32 results at the fp32
    tolerance; 16 (bf16 operands) is opt-in.  Returns (render_kwargs_train, render_kwargs_test, idx_iter, optimizer);
    in quirk mode the two dicts are the SAME object like upstream (:152)."""
    from ..rendering.render import render_rays, render_rays_eval
    is_use_dir = bool(args.use_viewdirs)
    embed_pos, ch_pos = embedding.get_embedder(args.multires, ref_quirks=ref_quirks)
    embed_dir, ch_dir = embedding.get_embedder(args.multires_views, ref_quirks=ref_quirks) if is_use_dir else (None, 0)
    output_ch = 5 if args.N_importance else 4
    query = NetworkQuery(embed_pos, embed_dir, args.netchunk, ref_quirks)
    mk = lambda d, w, s: NeRF(n_layers=d, width_layers=w, channel_input=ch_pos, channel_output=output_ch,
                              list_skip_connection_layers=[4], channel_input_views=ch_dir,
                              is_use_view_directions=is_use_dir, device=device, seed=s, precision=precision)
    model_coarse = mk(args.netdepth, args.netwidth, seed)
    model_fine = mk(args.netdepth_fine, args.netwidth_fine, None if seed is None else seed + 1) if args.N_importance > 0 else None
    optimizer = Adam(learning_rate=args.lrate, betas=(0.9, 0.999), shared_state=ref_quirks)
    kw = {
        "use_viewdirs": is_use_dir, "white_bkgd": args.white_bkgd, "network_query_fn": query, "is_test": True,
        "render_rays_func": render_rays, "network_coarse": model_coarse, "n_depth_samples": args.n_depth_samples,
        "network_fine": model_fine, "perturb": args.perturb, "raw_noise_std": args.raw_noise_std,
        "N_importance": args.N_importance,
    }
    if args.dataset_type != "llff" or args.no_ndc:
        kw["ndc"] = False
        kw["lindisp"] = args.lindisp
    kw_test = kw if ref_quirks else dict(kw)
    kw_test["perturb"] = False
    kw_test["raw_noise_std"] = 0
    kw_test["is_test"] = False
    kw_test["render_rays_func"] = render_rays_eval
    return kw, kw_test, 0, optimizer
