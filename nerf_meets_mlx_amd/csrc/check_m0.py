#!/usr/bin/env python3
"""Scan gfx950 ISA for uses of M0 other than the LDS-DMA statements of csrc/mlp_ring.h: `s_mov_b32 m0, <sgpr>` directly in
front of a `global_load_lds_*` (the M0-clobbering form of the ring kernels) and the save / restore pair around it
(`s_mov_b32 <sgpr>, m0` ... DMA ... `s_mov_b32 m0, <sgpr>`: dma_frag / dma_frag_nt of the weight-gradient kernels).  The
split-precision translation units are built with NERF_DMA_CLOBBER_M0: the ring's asm writes M0 and does not restore it, which is
valid only while nothing the COMPILER generated reads M0 or expects a value it put there -- in ANY kernel of the unit (without
--kernels every kernel is scanned).
    python check_m0.py build/mlp22-hip-amdgcn-amd-amdhsa-gfx950.s [--kernels substr ...] [--no-scratch]"""
import argparse
import re
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("--kernels", nargs="*", default=[])
    ap.add_argument("--no-scratch", action="store_true", help="also fail when a scanned kernel contains scratch (spill) instructions: "
                    "the one-wave-per-SIMD kernels are sized to the 512-register file; a change that tips hipcc into scratch costs "
                    "them a factor (round 4: +3 ms on the split-bf16 chain kernel) without failing any test")
    ap.add_argument("--allow-scratch", nargs=2, action="append", default=[], metavar=("KERNEL_SUBSTRING", "MAX"),
                    help="a kernel that may hold up to MAX scratch instructions (an opt-in, experimental variant whose spills were looked at)")
    a = ap.parse_args()
    text = open(a.asm).read()
    bad, seen = 0, 0
    for m in re.finditer(r"^(_Z[^\n:]*):[^\n]*\n", text, re.M):
        name = m.group(1)
        if a.kernels and not any(k in name for k in a.kernels):
            continue
        end = text.find(".Lfunc_end", m.end())
        nxt = re.search(r"^_Z[^\n:]*:", text[m.end():], re.M)
        if end < 0 or (nxt and m.end() + nxt.start() < end):
            continue                                   # a data symbol (a __device__ variable): no function end before the next symbol
        body = [ln.split(";")[0].strip() for ln in text[m.end():end].split("\n")]
        body = [ln for ln in body if ln and not ln.startswith(".")]
        if not body:
            continue                                   # a data symbol (a __device__ variable), not a function
        seen += 1
        ours = other = 0
        for i, ln in enumerate(body):
            if not re.search(r"\bm0\b", ln):
                continue
            nxt = [x for x in body[i + 1:i + 4]]
            prv = [x for x in body[max(0, i - 3):i]]
            if re.match(r"s_mov_b32 m0, s\d+$", ln) and any(x.startswith("global_load_lds") for x in nxt):
                ours += 1
            elif re.match(r"s_mov_b32 s\d+, m0$", ln) and any(x.startswith("global_load_lds") for x in body[i + 1:i + 5]):
                pass                                   # save in front of a DMA statement (dma_frag / dma_frag_nt)
            elif re.match(r"s_mov_b32 m0, s\d+$", ln) and any(x.startswith("global_load_lds") for x in prv):
                pass                                   # ... and its restore behind it
            else:
                other += 1
                if other <= 5:
                    print(f"    {name}: compiler-side M0 use: {ln}")
        spills = sum(1 for ln in body if ln.startswith("scratch_")) if a.no_scratch else 0
        print(f"{name}: {ours} LDS-DMA M0 writes, {other} other M0 uses" + (f", {spills} scratch instructions" if a.no_scratch else ""))
        allowed = max([int(n) for k, n in a.allow_scratch if k in name] or [0])
        bad += other + (spills if spills > allowed else 0)
    if not seen:
        print("no kernel matched", file=sys.stderr)
        return 2
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
