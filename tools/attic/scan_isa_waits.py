#!/usr/bin/env python3
"""Per kernel of a gfx950 ISA listing (hipcc -S --cuda-device-only): global loads, stores, `s_waitcnt vmcnt` statements and how
many of those are vmcnt(0) -- a quick way to spot load -> wait -> use -> store chains that the compiler could not batch
(stores that may alias later loads keep them in program order: the fp32 chain kernel's start-up spent 64 serialized L2 round
trips that way).  A kernel whose wait count approaches its load count with mostly small counts deserves a look.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -S --cuda-device-only -o /tmp/k.s nerf_meets_mlx_amd/csrc/composite.hip
    python tools/scan_isa_waits.py /tmp/k.s [--min-loads 12]
"""
import argparse
import re


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm", nargs="+")
    ap.add_argument("--min-loads", type=int, default=12)
    a = ap.parse_args()
    print("file,kernel,global_loads,global_stores,vmcnt_waits,vmcnt0_waits,small_count_waits")
    for path in a.asm:
        s = open(path).read()
        for m in re.finditer(r"^(_Z[^\n:]*):[^\n]*\n", s, re.M):
            j = s.find(".end_amdhsa_kernel", m.end())
            if j < 0:
                continue
            body = s[m.end():j].split("\n")
            loads = sum(1 for ln in body if re.match(r"\s+(global|buffer|flat)_load", ln) and "lds" not in ln)
            stores = sum(1 for ln in body if re.match(r"\s+(global|buffer|flat)_store", ln))
            waits = [int(x) for ln in body for x in re.findall(r"vmcnt\((\d+)\)", ln) if "s_waitcnt" in ln]
            if loads >= a.min_loads:
                print(f"{path.split('/')[-1]},{m.group(1)[:80]},{loads},{stores},{len(waits)},{sum(1 for w in waits if w == 0)},"
                      f"{sum(1 for w in waits if w <= 3)}")


if __name__ == "__main__":
    main()
