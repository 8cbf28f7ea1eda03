"""In-kernel clock and MFMA cycle utilisation of the fused render-path forward (diagnostic build only).

    make -C nerf_meets_mlx_amd/csrc stamp
    NERF_HIP_LIB=tools/diag/libnerf_stamp.so python tools/probe_clock.py

clock = d(s_memtime) / d(s_memrealtime) x 100 MHz, stamped around the persistent loop of every workgroup after
>= 2 s of back-to-back launches on random data (MI355X_MICROARCH.md, DVFS item 6)."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from nerf_meets_mlx_amd import _native
from nerf_meets_mlx_amd.models.NeRF import NeRF
dev = "cuda"
lib = _native.lib()
lib.nerf_debug_stamps.restype = C.c_int
lib.nerf_debug_stamps.argtypes = [C.c_void_p, C.c_int]
torch.manual_seed(0)
B, n = 32768, 192
o = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1) * 4
d = -o / 4 + 0.2 * torch.randn(B, 3, device=dev)
r = torch.zeros(B, 11, device=dev); r[:, :3] = o; r[:, 3:6] = d; r[:, 6] = 2; r[:, 7] = 6
r[:, 8:] = d / d.norm(dim=-1, keepdim=True)
z = torch.sort(torch.rand(B, n, device=dev) * 4 + 2, -1).values
m = NeRF(channel_input=63, channel_input_views=27, is_use_view_directions=True, device=dev, seed=0)
m.load_flat(m.params * 1.5)          # O(1) activations through all layers: random, non-trivial MFMA operands
for variant, mfma_per_wave, cyc, wps in ((3, 1159, 32, 2), (4, 1172 * 2, 16, 2), (5, 1172 * 4, 16, 1)):
    _native.check(lib.nerf_set_option(b"mlp_variant", variant))
    t0 = time.time()
    while time.time() - t0 < 2.5:
        for _ in range(20): m.query(r, z)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); m.query(r, z); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    nwg = min(512, (B * n + 255) // 256)
    st = np.zeros((nwg, 4), dtype=np.uint64)
    assert lib.nerf_debug_stamps(st.ctypes.data, nwg) == 0
    st = st[st[:, 3] == 1].astype(np.float64)
    ghz = st[:, 0] / st[:, 1] * 0.1
    cyc_per_pass = st[:, 0] / st[:, 2]
    # per SIMD: 2 waves, each issuing mfma_per_wave MFMAs of `cyc` cycles per pass (8 tiles of 32 samples per WG)
    busy = wps * mfma_per_wave * cyc
    print(f"variant {variant}: {ms:.3f} ms {2*593408*B*n/ms/1e9:.0f} TFLOP/s | in-kernel clock median {np.median(ghz):.3f} GHz "
          f"(min {ghz.min():.3f} max {ghz.max():.3f}) | cycles/pass median {np.median(cyc_per_pass):.0f}, MFMA-busy {busy} "
          f"= {busy/np.median(cyc_per_pass)*100:.1f} % of cycles | peak at this clock "
          f"{2.5e3*np.median(ghz)/2.4:.0f} TFLOP/s", flush=True)
