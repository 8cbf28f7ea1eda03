#!/usr/bin/env python3
"""Shim (command line only): the scanner lives next to the Makefile that runs it, nerf_meets_mlx_amd/csrc/check_m0.py, so that
the library builds from the package directory alone."""
import os, runpy, sys
if __name__ == "__main__":
    runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nerf_meets_mlx_amd", "csrc", "check_m0.py"),
                   run_name="__main__")
