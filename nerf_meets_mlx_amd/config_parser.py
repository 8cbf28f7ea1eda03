"""Flag set / text-config loader with the reference's names and defaults.

Boundary mirror of `mlx_nerf/config_parser.py:3-122`: same flag names,
types, defaults and help strings so that reference configs and scripts keep working;
pinned by tests/golden/ref_config_defaults.json (made from the reference's own parser).
The flag table is data; `config_parser()` builds the argparse object from it.
"""
import argparse
from typing import Dict, Optional

_S, _I, _F = str, int, float
_FLAG = "store_true"

# (name, kind, default, help).  kind is a type or _FLAG.
_FLAGS = [
    ("expname", _S, None, "experiment name"),
    ("basedir", _S, "./logs/", "where to store checkpoints and logs"),
    ("datadir", _S, "./data/llff/fern", "input data directory"),
    # training
    ("netdepth", _I, 8, "layers in network"),
    ("netwidth", _I, 256, "channels per layer"),
    ("netdepth_fine", _I, 8, "layers in fine network"),
    ("netwidth_fine", _I, 256, "channels per layer in fine network"),
    ("N_rand", _I, 32 * 32 * 4, "batch size (number of random rays per gradient step)"),
    ("lrate", _F, 5e-4, "learning rate"),
    ("lrate_decay", _I, 250, "exponential learning rate decay (in 1000 steps)"),
    ("chunk", _I, 1024 * 32, "number of rays processed in parallel"),
    ("netchunk", _I, 1024 * 64, "number of points sent through network in parallel"),
    ("no_batching", _FLAG, False, "only take random rays from 1 image at a time"),
    ("no_reload", _FLAG, False, "do not reload weights from saved checkpoint"),
    ("ft_path", _S, None, "specific weights file to reload for coarse network"),
    ("precrop_iters", _I, 0, "number of steps to train on central crops"),
    ("precrop_frac", _F, 0.5, "fraction of image taken for central crops"),
    # rendering
    ("n_depth_samples", _I, 64, "number of coarse samples per ray"),
    ("N_importance", _I, 0, "number of additional fine samples per ray"),
    ("perturb", _F, 1.0, "set to 0. for no jitter, 1. for jitter"),
    ("use_viewdirs", _FLAG, False, "use full 5D input instead of 3D"),
    ("i_embed", _I, 0, "set 0 for default positional encoding, -1 for none"),
    ("multires", _I, 10, "log2 of max freq for positional encoding (3D location)"),
    ("multires_views", _I, 4, "log2 of max freq for positional encoding (2D direction)"),
    ("raw_noise_std", _F, 0.0, "std dev of noise added to regularize sigma_a output"),
    ("render_only", _FLAG, False, "do not optimize, reload weights and render out render_poses path"),
    ("render_test", _FLAG, False, "render the test set instead of render_poses path"),
    ("render_factor", _I, 0, "downsampling factor to speed up rendering"),
    # dataset
    ("dataset_type", _S, "llff", "options: llff / blender / deepvoxels"),
    ("testskip", _I, 8, "will load 1/N images from test/val sets"),
    ("shape", _S, "greek", "options: armchair / cube / greek / vase"),
    ("white_bkgd", _FLAG, False, "render synthetic data on a white background"),
    ("half_res", _FLAG, False, "load blender synthetic data at 400x400 instead of 800x800"),
    ("factor", _I, 8, "downsample factor for LLFF images"),
    ("no_ndc", _FLAG, False, "do not use normalized device coordinates"),
    ("lindisp", _FLAG, False, "sampling linearly in disparity rather than depth"),
    ("spherify", _FLAG, False, "set for spherical 360deg scenes"),
    ("llffhold", _I, 8, "will take every 1/N images as LLFF test set"),
    # logging
    ("i_print", _I, 100, "frequency of console printout and metric logging"),
    ("i_img", _I, 500, "frequency of image logging"),
    ("i_weights", _I, 10000, "frequency of weight checkpoint saving"),
    ("i_testset", _I, 50000, "frequency of testset saving"),
    ("i_video", _I, 50000, "frequency of render_poses video saving"),
]


def config_parser() -> argparse.ArgumentParser:
    """Same flags/defaults as `mlx_nerf/config_parser.py:3-80`."""
    parser = argparse.ArgumentParser()
    for name, kind, default, doc in _FLAGS:
        if kind is _FLAG:
            parser.add_argument(f"--{name}", action="store_true", help=doc)
        else:
            parser.add_argument(f"--{name}", type=kind, default=default, help=doc)
    return parser


def load_config(parser: Optional[argparse.ArgumentParser], filename_config: str = "configs/lego.txt") -> Dict[str, str]:
    """`key = value` text file -> dict of STRINGS (`mlx_nerf/config_parser.py:82-101`).
    `parser` is accepted and ignored, as in the reference."""
    configs: Dict[str, str] = {}
    with open(filename_config, "r") as fp:
        for raw in fp:
            line = raw.strip()
            if not line:
                continue
            parts = line.split(" = ")
            configs[parts[0]] = parts[1]
    return configs


def _as_bool(v) -> bool:
    return v if isinstance(v, bool) else str(v).strip().lower() in ("1", "true", "yes", "on")


_COPIED = (  # (args attribute, config key, converter) -- config_parser.py:106-119
    ("expname", "expname", str), ("basedir", "basedir", str), ("datadir", "datadir", str),
    ("dataset_type", "dataset_type", str), ("no_batching", "no_batching", None),
    ("use_viewdirs", "use_viewdirs", None), ("white_bkgd", "white_bkgd", None),
    ("lrate_decay", "lrate_decay", int), ("n_depth_samples", "N_samples", int),
    ("N_importance", "N_importance", int), ("N_rand", "N_rand", int),
    ("precrop_iters", "precrop_iters", int), ("precrop_frac", "precrop_frac", float),
    ("half_res", "half_res", None),
)


def update_NeRF_args(args: argparse.Namespace, configs: dict, ref_quirks: bool = True) -> argparse.Namespace:
    """Copy the 14 config keys onto `args` and force `no_reload` (config_parser.py:104-122).
    ref_quirks=True keeps the four boolean keys as the file's STRINGS (any non-empty
    string is truthy: SURVEY Q2); ref_quirks=False parses them as booleans."""
    for attr, key, conv in _COPIED:
        val = configs[key]
        if conv is None:
            val = val if ref_quirks else _as_bool(val)
        else:
            val = conv(val)
        setattr(args, attr, val)
    args.no_reload = True
    return args
