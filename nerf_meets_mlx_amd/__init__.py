"""MI355X-native NeRF volumetric renderer: drop-in for the `mlx_nerf` hot path.

Host code (this package) holds tensors in PyTorch-ROCm and calls hand-written HIP
kernels (csrc/, gfx950) through the C ABI declared in include/nerf_hip.h.
"""
__version__ = "0.1.0"
