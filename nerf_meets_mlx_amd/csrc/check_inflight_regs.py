#!/usr/bin/env python3
"""Scan gfx950 ISA (hipcc -S --cuda-device-only) for reads of a register that an asynchronous vector-memory load may
still be writing.

Why: the fp32 chain / dW kernels (csrc/mlp32.hip) issue their operand loads with `asm volatile` and wait for them with
counted `s_waitcnt vmcnt(N)` asm statements of their own.  hipcc treats the asm's output registers as defined the moment
the asm statement has executed, so it is free to copy them (v_accvgpr_write to park them in AGPRs, v_mov for a live-range
split) or to use them as an address BEFORE the data has arrived -- which it does as soon as more values are live than the
256 VGPRs hold.  Nothing in the compiler or at run time reports that; the results are stale values in a few lanes.

What it does: a linear scan per kernel (control flow is ignored: loops and branches are walked in text order).  Every
global/buffer load instruction joins an in-order list with its destination registers (VGPRs or AGPRs); `s_waitcnt vmcnt(N)`
retires all but the N youngest loads (stores are ignored, which makes the model conservative: a store in the counter can
only mean that MORE loads have completed); any instruction that reads or overwrites a register of a load still on the
list is reported.  Kernels whose loads are all compiler-managed never trip it in straight-line code; loops can produce
false positives there (pass --kernels to restrict the scan to the asm-load kernels).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -S --cuda-device-only -o /tmp/mlp32.s nerf_meets_mlx_amd/csrc/mlp32.hip
    python nerf_meets_mlx_amd/csrc/check_inflight_regs.py /tmp/mlp32.s --kernels mlp32_fwd_kernel mlp32_bwd_kernel mlp32_dw_kernel
"""
import argparse
import re
import sys


def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"([va])\[(\d+):(\d+)\]$", tok)
    if m:
        return {(m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"([va])(\d+)$", tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    return set()


def kernels(text):
    for m in re.finditer(r"^(_Z[^\n:]*):[^\n]*\n", text, re.M):
        end = text.find(".end_amdhsa_kernel", m.end())
        nxt = re.search(r"^_Z[^\n:]*:", text[m.end():], re.M)
        stop = len(text) if end < 0 else end
        if nxt and m.end() + nxt.start() < stop:
            stop = m.end() + nxt.start()
        yield m.group(1), text[m.end():stop].split("\n")


def scan(body, verbose=0):
    loads, bad, n_loads = [], [], 0
    for ln, raw in enumerate(body):
        t = raw.split(";")[0].strip()
        if not t or t.startswith("."):
            continue
        parts = t.replace(",", " ").split()
        op = parts[0]
        pend = set().union(*loads) if loads else set()
        # a RETURNING atomic (sc0) is a load for this purpose: its destination VGPR is written when the data comes back
        # (the pass-queue ticket of the ring kernels, mlp_ring.h)
        if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")) or (op.startswith(("global_atomic", "buffer_atomic", "flat_atomic")) and " sc0" in t):
            if "lds" in t:                       # LDS-DMA: counted by vmcnt, no register destination
                loads.append(set())
                continue
            if any(regs(tok) & pend for tok in parts[2:]):
                bad.append((ln, "address from a pending register", t))
            loads.append(regs(parts[1]))
            n_loads += 1
            continue
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", t)
            if m:
                n = int(m.group(1))
                loads = loads[len(loads) - n:] if n > 0 else []
            continue
        is_store = op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "ds_write", "global_atomic"))
        srcs = parts[1:] if is_store else parts[2:]
        if any(regs(tok) & pend for tok in srcs):
            bad.append((ln, "reads a pending register", t))
        if not is_store and len(parts) > 1 and regs(parts[1]) & pend:
            bad.append((ln, "overwrites a pending register", t))
    return n_loads, bad


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("asm")
    ap.add_argument("--kernels", nargs="*", default=[], help="substrings of the (mangled) kernel names to scan; default: all")
    ap.add_argument("--show", type=int, default=5)
    a = ap.parse_args()
    text = open(a.asm).read()
    total, seen = 0, 0
    for name, body in kernels(text):
        if a.kernels and not any(k in name for k in a.kernels):
            continue
        seen += 1
        n_loads, bad = scan(body)
        print(f"{name}: {n_loads} loads, {len(bad)} suspicious")
        for ln, what, t in bad[:a.show]:
            print(f"    line {ln}: {what}: {t}")
        total += len(bad)
    if not seen:
        print("no kernel matched", file=sys.stderr)
        return 2
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
