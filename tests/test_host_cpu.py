"""CPU suite for the host side: the C-ABI library loads and exports every symbol that
include/nerf_hip.h declares (no compute calls without a GPU), plus host-only logic."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "nerf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nerf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    from nerf_meets_mlx_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        g.build()
    lib = _native.lib()
    syms = _declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in nerf_hip.h but not exported"
        assert s in _native.SIGNATURES, f"{s} has no ctypes signature"
    assert set(_native.SIGNATURES) == set(syms)
    assert lib.nerf_abi_version() == 3                       # host-only call, no GPU needed


def test_host_argument_validation_without_gpu():
    import ctypes as C
    from nerf_meets_mlx_amd import _native
    lib = _native.lib()
    assert lib.nerf_sh_encode(None, 4, 2, None, None) == -1                  # NERF_E_NULL
    assert b"NULL" in lib.nerf_last_error()
    assert lib.nerf_importance_sample(C.c_void_p(8), C.c_void_p(8), C.c_void_p(8), 4, 300, 8, 1e-5, None, None, None,
                                      None, None) == -2                      # NERF_E_SHAPE
    arch = _native.MlpArch(8, 256, 63, 27, 4, 1, 4, 16)
    arch32 = _native.MlpArch(8, 256, 63, 27, 4, 1, 4, 32)               # ABI 3: precision is a field of the arch
    assert lib.nerf_mlp_param_count(C.byref(arch)) == 595844 == lib.nerf_mlp_param_count(C.byref(arch32))
    bf16_image = (1184 + 1120) * 1024 + 2496 * 4 + 1184 * 1024          # fwd + bwd streams, biases, 16x16x32 stream
    fp32_image = (580 + 544) * 4096 + 3584 * 4                           # fp32 reference-precision streams + tail (round 5: + the image model's 4 x 256 output head)
    assert lib.nerf_mlp_packed_bytes(C.byref(arch)) == bf16_image
    assert lib.nerf_mlp_packed_bytes(C.byref(_native.MlpArch(8, 256, 63, 27, 4, 1, 4, 0))) == bf16_image     # 0 = default = 16
    assert lib.nerf_mlp_packed_bytes(C.byref(arch32)) == bf16_image + fp32_image
    arch22 = _native.MlpArch(8, 256, 63, 27, 4, 1, 4, 22)               # float32 tolerance on the 16-bit matrix pipe (round 4)
    f22_image = 2368 * 1024 + 2496 * 4                                   # split-fp16 (hi, lo) forward stream + bias slots
    s16_image = (2368 + 2208) * 1024                                     # split-bf16 (hi, lo) forward + transposed streams
    assert lib.nerf_mlp_packed_bytes(C.byref(arch22)) == bf16_image + f22_image + s16_image
    assert lib.nerf_mlp_acts_bytes(C.byref(arch22), 65) == 8 * 325 * 1024        # hi blocks | lo blocks | sign-bit words
    assert lib.nerf_mlp_dz_bytes(C.byref(arch22), 65) == 8 * 308 * 1024 + 512 * (64 * 1024 + 256) * 4
    assert lib.nerf_mlp_acts_bytes(C.byref(arch), 65) == 8 * 167 * 1024          # 3 tiles, padded to a whole 8-tile super-tile
    assert lib.nerf_mlp_acts_bytes(C.byref(arch32), 65) == 3 * 2592 * 128         # fp32 stores: rows x 32 floats per tile (2528 activation rows + 64 of ReLU sign bits)
    assert lib.nerf_mlp_dz_bytes(C.byref(arch32), 65) == 3 * 2496 * 128 + 2048 * (4096 + 64) * 4      # + split-K partial blocks of dW
    assert lib.nerf_get_option(b"mlp_precision") == -2 ** 31 and lib.nerf_get_option(b"nonsense") == -2 ** 31
    assert lib.nerf_set_option(b"mlp_precision", 32) == -3                        # gone: NERF_E_UNSUPPORTED, with a pointer to the arch
    assert b"nerf_mlp_arch.precision" in lib.nerf_last_error()
    img = _native.MlpArch(8, 256, 40, 0, 4, 0, 3, 16)
    assert lib.nerf_mlp_param_count(C.byref(img)) == 482051
    bad = _native.MlpArch(8, 128, 63, 27, 4, 1, 4, 16)
    assert lib.nerf_mlp_param_count(C.byref(bad)) == -1
    assert lib.nerf_mlp_param_count(C.byref(_native.MlpArch(8, 256, 63, 27, 4, 1, 4, 24))) == -1       # unknown precision
    # round 5: the image-fitting and the 2 x 64 models have reference-tolerance kernels too (split bf16; image: fp32 MFMA as well)
    img_bf16 = (960 + 928) * 1024 + 2080 * 4
    assert lib.nerf_mlp_packed_bytes(C.byref(img)) == img_bf16
    assert lib.nerf_mlp_packed_bytes(C.byref(_native.MlpArch(8, 256, 40, 0, 4, 0, 3, 22))) == img_bf16 + (1920 + 1856) * 1024
    assert lib.nerf_mlp_packed_bytes(C.byref(_native.MlpArch(8, 256, 40, 0, 4, 0, 3, 32))) == img_bf16 + fp32_image
    assert lib.nerf_mlp_acts_bytes(C.byref(_native.MlpArch(8, 256, 40, 0, 4, 0, 3, 22)), 65) == 8 * 270 * 1024
    assert lib.nerf_mlp_acts_bytes(C.byref(_native.MlpArch(8, 256, 40, 0, 4, 0, 3, 32)), 65) == 3 * 2592 * 128
    ngp = _native.MlpArch(2, 64, 32, 16, -1, 1, 4, 16)
    ngp22 = _native.MlpArch(2, 64, 32, 16, -1, 1, 4, 22)
    assert lib.nerf_mlp_param_count(C.byref(ngp22)) == 13188
    assert lib.nerf_mlp_packed_bytes(C.byref(ngp22)) == lib.nerf_mlp_packed_bytes(C.byref(ngp)) + 128 * 1024
    assert lib.nerf_mlp_acts_bytes(C.byref(ngp22), 65) == 8 * 37 * 1024 and lib.nerf_mlp_dz_bytes(C.byref(ngp22), 65) == 8 * 32 * 1024 + 512 * (64 * 1024 + 256) * 4
    assert lib.nerf_mlp_packed_bytes(C.byref(_native.MlpArch(2, 64, 32, 16, -1, 1, 4, 32))) == -1      # fp32 MFMA: the 8 x 256 models only
    with pytest.raises(ValueError):
        _native.ptr(torch.zeros(3))                                          # CPU tensors are refused: no fallback


def test_pixel_permutation_host_properties():
    from nerf_meets_mlx_amd.ops.index import pixel_permutation_host
    full = pixel_permutation_host(4096, 4096, 11)
    assert sorted(full.tolist()) == list(range(4096))
    a = pixel_permutation_host(1024, 640000, 5, 0)
    b = pixel_permutation_host(1024, 640000, 5, 1024)
    assert len(set(a.tolist()) | set(b.tolist())) == 2048                    # consecutive batches never repeat a pixel
    assert not np.array_equal(a, pixel_permutation_host(1024, 640000, 6, 0))
    # roughly uniform over the image
    assert abs(a.mean() / 640000 - 0.5) < 0.05


def test_pose_and_layout_match_oracle():
    from nerf_meets_mlx_amd.ops.pose import pose_spherical
    from nerf_meets_mlx_amd.models.NeRF import layer_shapes
    for th in (-180.0, -90.0, 0.0, 33.0):
        assert torch.equal(pose_spherical(th, -30.0, 4.0), O.pose_spherical(th, -30.0, 4.0))
    assert layer_shapes() == O.NerfArch().layer_shapes()
    assert layer_shapes(cin=40, cdir=0, use_viewdirs=False, cout=3) == O.NerfArch(40, 0, 3, use_viewdirs=False).layer_shapes()


def test_product_package_never_imports_oracle():
    import subprocess
    import sys
    out = subprocess.run(["grep", "-rIl", "-E", r"^\s*(from|import) +oracle", os.path.join(ROOT, "nerf_meets_mlx_amd")],
                         capture_output=True, text=True).stdout.strip()
    assert out == "", f"product path imports the oracle: {out}"


def test_blender_loader_roundtrip(tmp_path):
    """dataset/dataloader.py:20-111 on a tiny generated dataset (8x8 RGBA PNGs)."""
    import json
    from PIL import Image
    from nerf_meets_mlx_amd.dataset.dataloader import load_blender_data, post_load_blender_data
    rng = np.random.default_rng(0)
    counts = {"train": 3, "val": 2, "test": 4}
    for split, n in counts.items():
        os.makedirs(tmp_path / split, exist_ok=True)
        frames = []
        for i in range(n):
            arr = rng.integers(0, 256, size=(8, 8, 4), dtype=np.uint8)
            Image.fromarray(arr, "RGBA").save(tmp_path / split / f"r_{i}.png")
            frames.append({"file_path": f"./{split}/r_{i}", "transform_matrix": np.eye(4).tolist()})
        with open(tmp_path / f"transforms_{split}.json", "w") as fp:
            json.dump({"camera_angle_x": 0.6911112070083618, "frames": frames}, fp)
    imgs, poses, rposes, (H, W, f), i_split = load_blender_data(str(tmp_path), testskip=2)
    assert imgs.shape == (3 + 1 + 2, 8, 8, 4) and imgs.dtype == np.float32 and poses.shape == (6, 4, 4)
    assert [len(s) for s in i_split] == [3, 1, 2] and rposes.shape == (160, 4, 4)
    assert abs(f - 0.5 * 8 / np.tan(0.5 * 0.6911112070083618)) < 1e-9
    first = np.asarray(Image.open(tmp_path / "train" / "r_0.png")) / 255.0
    np.testing.assert_allclose(imgs[0], first.astype(np.float32))
    i_train, i_val, i_test, near, far, rgb = post_load_blender_data(i_split, imgs, True)
    np.testing.assert_allclose(rgb, imgs[..., :3] * imgs[..., 3:] + (1 - imgs[..., 3:]))
    assert (near, far) == (2.0, 6.0)
    half = load_blender_data(str(tmp_path), half_res=True, testskip=1)
    assert half[0].shape[1:3] == (4, 4) and abs(half[3][2] - f / 2) < 1e-9


def test_bench_and_entry_scripts_parse_without_a_gpu():
    """bench.py --help and importing __graft_entry__ must work on a CPU-only box (the driver's build check runs there)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-1000:]
    for flag in ("--gpus", "--steps", "--warmup", "--config", "--n-importance"):
        assert flag in out.stdout
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; assert callable(g.build) and callable(g.smoke)"],
                         capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stderr[-1000:]
    for tool in ("summarize_rocprof.py", "psnr_parity.py", "psnr_parity_ngp.py", "bench_kernels.py"):
        src = open(os.path.join(root, "tools", tool)).read()
        compile(src, tool, "exec")


def _load_bench():
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_self_spawns_ranks_and_refuses_a_world_mismatch(monkeypatch):
    """`python bench.py --gpus N` with no launcher around it must start N ranks itself (torch.distributed.run, rendezvous
    on 127.0.0.1, dmabuf IPC flag kept) before anything touches the GPU -- in round 1 it silently ran ONE rank and printed
    n_gpus: 1 -- and a launcher that started a different number of ranks than --gpus says is an error, not a 1-GPU line."""
    import os, subprocess, sys
    b = _load_bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        b.main()
    assert e.value.code == 7                                       # the children's exit code is ours
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert os.path.basename(cmd[cmd.index("--master-port") + 2]) == "bench.py"
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # under a launcher whose world size differs from --gpus: exit 2 before any GPU work
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode == 2 and "WORLD_SIZE=1" in out.stderr


def test_bench_reads_pmc_tables_with_and_without_quotes(tmp_path):
    """Kernel names carry commas (`mlp_fwd_ring16_kernel<8, 2>`): round 1's unquoted CSV shifted the columns and the bench
    line lost `mfma_busy_cycles_frac`.  Both spellings must parse."""
    b = _load_bench()
    hdr = "kernel,grid,launches,avg_us,SQ_VALU_MFMA_BUSY_CYCLES,GRBM_GUI_ACTIVE,clock_GHz,mfma_busy_frac_of_cycles,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,lds_conflict_frac\n"
    row = "131072,6,4987.4,7373586432,73472174,1.841,0.784,0,1042861568,0.0000\n"
    other = "nerf::mlp_dw_kernel,262144,6,1566.8,956301312,25121947,2.004,0.297,0,119544580,0.0000\n"
    for name in ("nerf::mlp_fwd_ring16_kernel<8, 2>,", '"nerf::mlp_fwd_ring16_kernel<8, 2>",'):
        f = tmp_path / "t.csv"
        f.write_text(hdr + other + name + row)
        assert b._read_pmc_busy(str(f), "mlp_fwd_ring16_kernel<8,2>") == 0.784
        assert b._read_pmc_busy(str(f), "mlp_dw_kernel") == 0.297
        assert b._read_pmc_busy(str(f), "no_such_kernel") is None
    assert b._read_pmc_busy(str(tmp_path / "absent.csv"), "x") is None


def test_counter_seed_streams_are_disjoint_and_stateless():
    """Trainer randomness is a function of (seed, rank, stream, iteration): no RNG state in checkpoints, and a resumed
    rank != 0 keeps its own streams (rank 0's saved generators used to overwrite every rank's)."""
    from nerf_meets_mlx_amd import parallel
    seen = set()
    for rank in range(8):
        for stream in (1, 2, 3):
            for it in range(0, 2000, 7):
                s = parallel.counter_seed(4, rank, stream, it)
                assert 0 <= s < (1 << 63)
                seen.add(s)
    assert len(seen) == 8 * 3 * len(range(0, 2000, 7))                                 # no (rank, stream, it) collisions
    assert parallel.counter_seed(4, 1, 2, 10) == parallel.counter_seed(4, 1, 2, 10)    # stateless
    assert parallel.counter_seed(4, 0, 2, 7919) != parallel.counter_seed(4, 1, 2, 0)   # the old seed + it scheme collided here


def test_pixel_permutation_is_a_uniform_sampler():
    """The batch of a training iteration is `np.random.choice(H*W, N_rand, replace=False)` upstream
    (entrypoints/__test_nerf.py:229); here it is the first N_rand outputs of a keyed 4-round Feistel bijection of
    [0, H*W) (csrc/rays.hip perm_kernel; `pixel_permutation_host` is its bit-exact mirror, compared on the device in
    tests/test_gpu_parity.py).  Being a bijection is not enough to be a SAMPLER: over 10^4 keys the first 1024 of
    640 000 outputs must be uniform over the pixel range, over (row, col), in the low bits, with independent consecutive
    outputs and key-to-key overlap like independent draws -- each statistic next to the same statistic of numpy's choice,
    with chi-square bounds at mean + 5 sigma."""
    from nerf_meets_mlx_amd.ops.index import pixel_permutation_host
    dom, n, S = 640000, 1024, 10000
    A = np.stack([pixel_permutation_host(n, dom, 1000 + s, 0) for s in range(S)])
    rng = np.random.default_rng(1)
    B = np.stack([rng.choice(dom, n, replace=False) for _ in range(S)])

    def chi2(counts):
        e = counts.sum() / counts.size
        return float(((counts - e) ** 2 / e).sum())

    def bound(df):
        return df + 5.0 * np.sqrt(2.0 * df)

    for name, X in (("feistel", A), ("numpy", B)):
        assert all(len(np.unique(X[i])) == n for i in range(0, S, 500)), name                  # without replacement
        marg = chi2(np.bincount(X.reshape(-1) * 256 // dom, minlength=256))                     # uniform over the pixel list
        pairs = chi2(np.bincount(((X[:, :-1] * 16 // dom) * 16 + (X[:, 1:] * 16 // dom)).reshape(-1), minlength=256))   # consecutive outputs independent
        first = chi2(np.bincount(X[:, 0] * 32 // dom, minlength=32))                            # output 0 across keys
        rc = chi2(np.bincount((((X // 800) * 8 // 800) * 8 + ((X % 800) * 8 // 800)).reshape(-1), minlength=64))        # (row, col) blocks
        low = chi2(np.bincount(X.reshape(-1) & 255, minlength=256))                             # low index bits
        x = X.astype(np.float64) / dom - 0.5
        corr = float((x[:, :-1] * x[:, 1:]).mean() * 12.0)                                      # lag-1 serial correlation
        overlap = float(np.mean([len(np.intersect1d(X[i], X[i + 1])) for i in range(2000)]))    # E = 1024^2 / 640000 = 1.64
        assert marg < bound(255) and pairs < bound(255) and low < bound(255), (name, marg, pairs, low)
        assert first < bound(31) and rc < bound(63), (name, first, rc)
        assert abs(corr) < 5.0 / np.sqrt(S * (n - 1)) * 1.2, (name, corr)
        assert 1.3 < overlap < 2.0, (name, overlap)


def test_c_abi_header_compiles_as_c_and_library_links_from_c(tmp_path):
    """The boundary is a C ABI: include/nerf_hip.h compiled by gcc as C (-std=c99 -Wall -Werror), linked against the in-tree
    libnerf_hip.so, host-only entry points called from C (tests/abi/abi_check.c) -- no Python, no GPU."""
    import shutil
    import subprocess
    import __graft_entry__ as g
    from nerf_meets_mlx_amd import _native
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    if not os.path.exists(_native.LIB_PATH):
        g.build()
    libdir = os.path.dirname(os.path.abspath(_native.LIB_PATH))
    exe = str(tmp_path / "abi_check")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "abi", "abi_check.c"),
                    "-L", libdir, "-lnerf_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "abi_check ok" in out.stdout


def test_run_log_jsonl_format(tmp_path):
    """engine/runlog.py (the reference keeps loss.item() per iteration and plots it, __test_nerf.py:298-299,314-322): one JSON
    object per line, rank 0 only, appended across a resume; train records carry it / losses / PSNR / lr / whole-job rays/s
    (evaluation renders excluded from the rate window), non-finite values become null."""
    import json
    import time
    from nerf_meets_mlx_amd.engine import runlog
    d = str(tmp_path / "exp")
    log = runlog.RunLog(d, rank=0, world=2)
    log.run(n_rand=1024, precision=22, resumed_from=None)
    log.mark(0)
    time.sleep(0.02)
    r1 = log.train(100, 0.01, 0.001, 5e-4, 1024)
    assert abs(r1["psnr_coarse"] - 20.0) < 1e-9 and abs(r1["psnr_fine"] - 30.0) < 1e-9
    assert 0 < r1["rays_per_s"] <= 100 * 1024 * 2 / 0.02
    log.eval(100, 23.5, 7, 5.0)                                       # a 5-second "render" does not count as training time
    time.sleep(0.02)
    r2 = log.train(200, float("nan"), None, 4e-4, 1024)
    assert r2["loss_coarse"] is None and r2["loss_fine"] is None and r2["psnr_coarse"] is None
    assert r2["rays_per_s"] is None or r2["rays_per_s"] > 0           # window minus the eval pause is <= 0 here -> null
    # a second process-lifetime (resume) appends; other ranks write nothing
    log2 = runlog.RunLog(d, rank=0, world=2)
    log2.run(resumed_from="000200.npz")
    runlog.RunLog(d, rank=1, world=2).train(300, 1.0, 1.0, 1e-4, 1024)
    lines = [ln for ln in open(os.path.join(d, "log.jsonl")).read().splitlines() if ln]
    recs = [json.loads(ln) for ln in lines]                           # every line is strict JSON (no NaN literals)
    assert [r["kind"] for r in recs] == ["run", "train", "eval", "train", "run"] and recs == runlog.read(os.path.join(d, "log.jsonl"))
    assert all("NaN" not in ln and "Infinity" not in ln for ln in lines)
    assert set(recs[1]) == {"kind", "it", "loss_coarse", "loss_fine", "psnr_coarse", "psnr_fine", "lr", "rays_per_s", "elapsed_s"}
    assert recs[2] == {"kind": "eval", "it": 100, "psnr": 23.5, "view": 7, "seconds": 5.0}
    assert recs[0]["world_size"] == 2 and recs[4]["resumed_from"] == "000200.npz"


def test_every_documented_option_is_gettable_and_settable():
    """include/nerf_hip.h lists the keys of nerf_set_option; nerf_get_option must know every one of them (advisor, round 5: three keys
    read as -1, the same value as 'unknown' and as the legitimate 'automatic' of dw_unit_bias) and a get / set pair must restore a
    setting; unknown keys read NERF_OPTION_UNKNOWN (INT_MIN) and are refused by set."""
    from nerf_meets_mlx_amd import _native
    lib = _native.lib()
    text = open(os.path.join(ROOT, "include", "nerf_hip.h")).read()
    block = text[text.index("runtime selection of kernel variants"):text.index("int nerf_set_option")]
    keys = sorted(set(re.findall(r'^ \*   "([a-z0-9_]+)"', block, flags=re.M)))
    assert len(keys) >= 14 and {"pass_queue", "f22_tiles", "dw_narrow_first", "dw_unit_bias", "dw_private_tiles", "dw_ring_cap"} <= set(keys), keys
    for k in keys + ["tile_pad16", "bwd_stage", "dw_job_mask"]:                  # + the undocumented diagnostic knobs
        v = lib.nerf_get_option(k.encode())
        assert v != -2 ** 31, f"nerf_get_option does not know '{k}'"
        assert lib.nerf_set_option(k.encode(), v) == 0, k                      # writing back what was read changes nothing ...
        assert lib.nerf_get_option(k.encode()) == v, k                          # ... (dw_unit_bias: -1 = automatic stays -1)
    assert lib.nerf_get_option(b"no_such_key") == -2 ** 31 and lib.nerf_set_option(b"no_such_key", 1) == -3
    assert lib.nerf_get_option(b"pass_queue") == 1 and lib.nerf_get_option(b"f22_tiles") == 0 and lib.nerf_get_option(b"dw_unit_bias") == -1
