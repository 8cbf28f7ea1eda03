#!/usr/bin/env python3
"""Shim: the scanner lives next to the Makefile that runs it (nerf_meets_mlx_amd/csrc/check_inflight_regs.py)."""
import os, sys
import importlib.util
_p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nerf_meets_mlx_amd", "csrc", "check_inflight_regs.py")
_spec = importlib.util.spec_from_file_location("_csrc_check_inflight_regs", _p)
_mod = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(_mod)
globals().update({k: v for k, v in vars(_mod).items() if not k.startswith("__")})
if __name__ == "__main__":
    sys.exit(_mod.main())
