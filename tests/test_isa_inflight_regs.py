"""The fp32 kernels load their operands with asm loads and wait for them with their own counted s_waitcnt: hipcc must
not touch a destination register before that wait (it parks such registers in AGPRs right behind the load as soon as the
256 VGPRs run out -- measured on a version of the dW kernel with four VGPR sets in flight).  This test compiles
csrc/mlp32.hip to ISA and scans it (csrc/check_inflight_regs.py); CPU only (hipcc cross-compiles)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nerf_meets_mlx_amd", "csrc"))      # the scanners live next to the Makefile that runs them
import check_inflight_regs as chk                                   # noqa: E402

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def test_scanner_flags_a_read_ahead_of_the_wait_and_accepts_the_counted_form():
    bad = """
        global_load_dwordx4 v[4:7], v[0:1], off
        global_load_dwordx4 v[8:11], v[0:1], off offset:16
        v_accvgpr_write_b32 a3, v5
        s_waitcnt vmcnt(0)
    """.split("\n")
    n, found = chk.scan(bad)
    assert n == 2 and len(found) == 1 and "reads a pending register" in found[0][1]
    good = """
        global_load_dwordx4 v[4:7], v[0:1], off
        global_load_dwordx4 a[8:11], v[0:1], off offset:16
        s_waitcnt vmcnt(1)
        v_mfma_f32_32x32x2_f32 a[16:31], v4, v12, a[16:31]
        s_waitcnt vmcnt(0)
        v_mfma_f32_32x32x2_f32 a[16:31], a8, v12, a[16:31]
    """.split("\n")
    assert chk.scan(good)[1] == []
    late = """
        global_load_dwordx4 v[4:7], v[0:1], off
        global_load_dwordx4 a[8:11], v[0:1], off offset:16
        s_waitcnt vmcnt(1)
        v_mfma_f32_32x32x2_f32 a[16:31], a8, v12, a[16:31]
    """.split("\n")
    assert len(chk.scan(late)[1]) == 1


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_fp32_kernels_never_touch_a_register_ahead_of_its_wait(tmp_path):
    out = str(tmp_path / "mlp32.s")
    src = os.path.join(ROOT, "nerf_meets_mlx_amd", "csrc", "mlp32.hip")
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only",
                    "-o", out, src], check=True, timeout=900)
    text = open(out).read()
    seen = 0
    for name, body in chk.kernels(text):
        if not any(k in name for k in ("mlp32_fwd_kernel", "mlp32_bwd_kernel", "mlp32_dw_kernel")):
            continue
        seen += 1
        n_loads, bad = chk.scan(body)
        assert n_loads > 50, name
        assert bad == [], (name, bad[:3])
    assert seen == 7                   # forward (store / no store) x (view / image model), chain (view / image), dW


def test_m0_scanner_accepts_the_ring_statements_and_flags_compiler_uses(tmp_path):
    """csrc/check_m0.py (run by csrc/Makefile on the ISA of the split-precision kernels, which write M0 without restoring
    it): the ring's `s_mov_b32 m0, sN` + `global_load_lds` pairs pass, any other mention of M0 fails the scan."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("csrc_check_m0", os.path.join(ROOT, "nerf_meets_mlx_amd", "csrc", "check_m0.py"))
    check_m0 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(check_m0)
    ok = tmp_path / "ok.s"
    ok.write_text("_ZN4nerf3f2216mlp22_fwd_kernelILi1EEEvNS_7FwdArgsE: ; @k\n\ts_mov_b32 m0, s12\n\ts_nop 0\n"
                  "\tglobal_load_lds_dwordx4 v3, s[4:5]\n\tv_mfma_f32_16x16x32_f16 v[0:3], v[4:7], a[0:3], v[0:3]\n.Lfunc_end0:\n")
    bad = tmp_path / "bad.s"
    bad.write_text("_ZN4nerf3f2216mlp22_fwd_kernelILi1EEEvNS_7FwdArgsE: ; @k\n\ts_mov_b32 m0, s12\n\ts_nop 0\n"
                   "\tglobal_load_lds_dwordx4 v3, s[4:5]\n\ts_mov_b32 s9, m0\n\tv_movrels_b32 v1, v2\n.Lfunc_end0:\n")
    sys.argv = ["check_m0.py", str(ok), "--kernels", "fwd_kernel"]
    assert check_m0.main() == 0
    sys.argv = ["check_m0.py", str(bad), "--kernels", "fwd_kernel"]
    assert check_m0.main() == 1
    sys.argv = ["check_m0.py", str(ok), "--kernels", "no_such_kernel"]
    assert check_m0.main() == 2
    # the save / restore form of the weight-gradient kernels (dma_frag / dma_frag_nt) is ours too; without --kernels every kernel is scanned
    sr = tmp_path / "sr.s"
    sr.write_text("_ZN4nerf3s1613s16_dw_kernelILb1EEEvNS_6DwArgsE: ; @k\n\ts_mov_b32 s7, m0\n\ts_mov_b32 m0, s12\n\ts_nop 0\n"
                  "\tglobal_load_lds_dwordx4 v[2:3], off nt\n\ts_mov_b32 m0, s7\n.Lfunc_end0:\n")
    sys.argv = ["check_m0.py", str(sr)]
    assert check_m0.main() == 0


def test_shipped_split_kernels_passed_the_build_time_scans():
    """csrc/Makefile scans the -save-temps ISA of the objects it ships: mlp32.o for reads of in-flight load registers, mlp22.o and
    mlp_s16.o for compiler-side M0 uses and scratch instructions, and fails the build on a hit.  This reads the scan records of
    the build in the tree (conftest.py runs `make` first): they must exist for every scanned object and report zero findings."""
    import re
    build = os.path.join(ROOT, "nerf_meets_mlx_amd", "csrc", "build")
    if not os.path.isdir(build):
        pytest.skip("no in-tree build directory (library built elsewhere)")
    # (object, kernels scanned, of which ring kernels that issue LDS-DMA): mlp_s16x's 2 x 64 kernels keep their weights LDS-resident
    # every kernel of a unit is scanned (round 5): pack kernels and the weight-gradient kernels included
    # round 6: mlp22.o holds a fourth kernel, the 48-samples-per-wave forward, which sits at the edge of the 512-register file: up to 6
    # scratch instructions are accepted THERE (thread id kept for the end of the kernel, two address temporaries at the head of a pass --
    # none between the head and the end of a pass, where a scratch reload would drain the weight ring), nowhere else
    edge = {"mlp22_fwd_kernelILi1ELi3E": 6}
    for name, kernels, ring in (("mlp22_m0_scan.txt", 4, 3), ("mlp_s16_m0_scan.txt", 6, 5), ("mlp_dww_m0_scan.txt", 2, 2),
                                ("mlp_s16x_m0_scan.txt", 10, 3)):
        path = os.path.join(build, name)
        assert os.path.exists(path), f"{name} missing: the Makefile rule of the split kernels did not run"
        rows = [ln for ln in open(path) if "LDS-DMA M0 writes" in ln]
        assert len(rows) == kernels, rows
        with_dma = 0
        for ln in rows:
            m = re.search(r": (\d+) LDS-DMA M0 writes, (\d+) other M0 uses, (\d+) scratch instructions", ln)
            allowed = max([n for k, n in edge.items() if k in ln] or [0])
            assert m and int(m.group(2)) == 0 and int(m.group(3)) <= allowed, ln
            with_dma += int(m.group(1)) > 0
        assert with_dma == ring, rows
    scan32 = os.path.join(build, "mlp32_inflight_scan.txt")
    assert os.path.exists(scan32)
    rows = [ln for ln in open(scan32) if "suspicious" in ln]
    assert len(rows) == 7 and all(ln.rstrip().endswith(" 0 suspicious") for ln in rows), rows
