#!/usr/bin/env python3
"""Shim: the scanner lives next to the Makefile that runs it (nerf_meets_mlx_amd/csrc/check_m0.py), so that the library builds
from the package directory alone."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nerf_meets_mlx_amd", "csrc"))
if "check_m0" in sys.modules and sys.modules["check_m0"].__file__ == __file__:
    del sys.modules["check_m0"]
import importlib.util
_spec = importlib.util.spec_from_file_location("_csrc_check_m0", os.path.join(sys.path[0], "check_m0.py"))
_mod = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(_mod)
main = _mod.main
if __name__ == "__main__":
    sys.exit(main())
