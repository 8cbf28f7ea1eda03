"""ctypes binding of libnerf_hip.so (the C ABI declared in include/nerf_hip.h).

There is NO fallback: if the library is missing or a call fails this raises.  The
product path never touches `oracle/`.
"""
import ctypes as C
import os

import torch  # noqa: F401  (loads torch's libamdhip64.so first so the HIP runtime is shared)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NERF_HIP_LIB", os.path.join(_HERE, "libnerf_hip.so"))   # override: A/B builds of the same ABI

# every symbol of include/nerf_hip.h: name -> (restype, argtypes)
_P, _I64, _I, _F, _U64 = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_uint64
SIGNATURES = {
    "nerf_abi_version": (_I, []),
    "nerf_last_error": (C.c_char_p, []),
    "nerf_pixel_permutation": (_I, [_P, _I64, _I64, _U64, _U64, _P]),
    "nerf_ray_gen": (_I, [_P, _I64, _I, _I, C.POINTER(C.c_double), C.POINTER(C.c_float), _F, _F, _P, _P, _P]),
    "nerf_sample_batch": (_I, [_I64, _I, _I, _U64, _U64, C.POINTER(C.c_double), C.POINTER(C.c_float), _F, _F, _P, _P, _P, _P, _P]),
    "nerf_gather_rows": (_I, [_P, _I64, _P, _I64, _I, _P, _P]),
    "nerf_ndc_rays": (_I, [_P, _I64, _I, _I, _F, _F, _P]),
    "nerf_sample_coarse": (_I, [_P, _I64, _I, _I, _F, _P, _P, _P]),
    "nerf_add_noise_z": (_I, [_P, _P, _I64, _I, _F, _P, _P]),
    "nerf_importance_sample": (_I, [_P, _P, _P, _I64, _I, _I, _F, _P, _P, _P, _P, _P]),
    "nerf_encode_freq": (_I, [_P, _I64, _I, _I, _I, _P, _P]),
    "nerf_encode_sinusoidal": (_I, [_P, _I64, _I, _I, C.POINTER(C.c_float), _I, _P, _P]),
    "nerf_sh_encode": (_I, [_P, _I64, _I, _P, _P]),
    "nerf_hashgrid_forward": (_I, [_P, _I64, _P, _I, _I, _I, C.POINTER(C.c_int), _P, _P]),
    "nerf_hashgrid_backward": (_I, [_P, _I64, _P, _I, _I, _I, C.POINTER(C.c_int), _P, _P]),
    "nerf_ngp_encode": (_I, [_P, _P, _I64, _I, _P, _I, _I, _I, C.POINTER(C.c_int), _I, C.c_float, C.c_float, _P, _P, _P]),
    "nerf_ngp_query_fused": (_I, [_P, _P, _P, _P, _I64, _I, _P, _I, _I, _I, C.POINTER(C.c_int), _I, C.c_float, C.c_float, _P, _P, _P]),
    "nerf_ngp_query_fused_h": (_I, [_P, _P, _P, _P, _I64, _I, _P, _P, _I, _I, _I, C.POINTER(C.c_int), _I, C.c_float, C.c_float, _P, _P, _P]),
    "nerf_hashgrid_backward_rays": (_I, [_P, _P, _I64, _I, _P, _I, _I, _I, C.POINTER(C.c_int), C.c_float, C.c_float, _P, _P]),
    "nerf_hashgrid_backward_ex": (_I, [_P, _I64, _P, _I, _I, _I, C.POINTER(C.c_int), _I, _I, _I, _P, _P]),
    "nerf_hashgrid_backward_rays_ex": (_I, [_P, _P, _I64, _I, _P, _I, _I, _I, C.POINTER(C.c_int), C.c_float, C.c_float, _I, _I, _I, _P, _P]),
    "nerf_composite_forward": (_I, [_P, _P, _P, _I64, _I, _F, _P, _I, _P, _P, _P, _P, _P, _P]),
    "nerf_composite_backward": (_I, [_P, _P, _P, _I64, _I, _F, _P, _I, _P, _P, _P, _P, _P]),
    "nerf_mse_loss_grad": (_I, [_P, _P, _I64, _F, _P, _P, _P]),
    "nerf_composite_mse_backward": (_I, [_P, _P, _P, _I64, _I, _I, _P, _F, _P, _P, _P, _P]),
    "nerf_ssim_sums": (_I, [_P, _P, _I, _I, _I, _I, C.POINTER(C.c_float), _I, _F, _F, _P, _P]),
    "nerf_mlp_param_count": (_I64, [_P]),
    "nerf_mlp_packed_bytes": (_I64, [_P]),
    "nerf_mlp_pack": (_I, [_P, _P, _P, _P]),
    "nerf_mlp_acts_bytes": (_I64, [_P, _I64]),
    "nerf_mlp_dz_bytes": (_I64, [_P, _I64]),
    "nerf_mlp_forward": (_I, [_P, _P, _P, _I64, _P, _P]),
    "nerf_mlp_forward_train": (_I, [_P, _P, _P, _I64, _P, _P, _P]),
    "nerf_query_fused": (_I, [_P, _P, _P, _P, _I64, _I, _I, _P, _P, _P]),
    "nerf_mlp_backward": (_I, [_P, _P, _P, _P, _I64, _P, _P, _P]),
    "nerf_mlp_backward_inputs": (_I, [_P, _P, _P, _P, _I64, _P, _P, _P, _P]),
    "nerf_mlp_debug_width": (_I, [_P, _I, _I]),
    "nerf_mlp_debug_read": (_I, [_P, _P, _I, _I, _I64, _P, _P]),
    "nerf_render_workspace_bytes": (_I64, [_I64, _I, _I]),
    "nerf_render_rays_fused": (_I, [_P, _P, _P, _P, _I64, _I, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nerf_comm_unique_id": (_I, [C.c_char_p]),
    "nerf_comm_init": (_I, [C.POINTER(C.c_void_p), _I, _I, C.c_char_p]),
    "nerf_allreduce_grads": (_I, [_P, _P, _I64, _P]),
    "nerf_comm_destroy": (_I, [_P]),
    "nerf_set_option": (_I, [C.c_char_p, _I]),
    "nerf_get_option": (_I, [C.c_char_p]),
    "nerf_adam_step": (_I, [_P, _P, _P, _P, _I64, _F, _F, _F, _F, _I, _I, _F, _P]),
    "nerf_adam_step_ex": (_I, [_P, _P, _P, _P, _I64, _F, _F, _F, _F, _I, _I, _F, _I, _I, _P]),
    "nerf_adam_step_shadow": (_I, [_P, _P, _P, _P, _I64, _F, _F, _F, _F, _I, _I, _F, _I, _I, _P, _P]),
}


class MlpArch(C.Structure):
    """struct nerf_mlp_arch (include/nerf_hip.h)."""
    _fields_ = [("n_layers", _I), ("width", _I), ("in_pos", _I), ("in_dir", _I), ("skip_layer", _I),
                ("use_viewdirs", _I), ("out_ch", _I), ("precision", _I)]


class NerfHipError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libnerf_hip.so once; raise loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NerfHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc: int):
    if rc != 0:
        msg = lib().nerf_last_error().decode(errors="replace")
        if rc in (-1, -2, -3):
            raise ValueError(f"libnerf_hip error {rc}: {msg}")
        raise NerfHipError(f"libnerf_hip error {rc}: {msg}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Tensors must be contiguous CUDA/ROCm."""
    if t is None:
        return None
    if not t.is_cuda:
        raise ValueError("libnerf_hip kernels need tensors on the ROCm device (there is no CPU path)")
    if not t.is_contiguous():
        raise ValueError("libnerf_hip kernels need contiguous tensors")
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def f32(t, device=None):
    """float32 contiguous view/copy on the device."""
    if not torch.is_tensor(t):
        t = torch.as_tensor(t)
    if device is not None:
        t = t.to(device)
    return t.to(torch.float32).contiguous()
