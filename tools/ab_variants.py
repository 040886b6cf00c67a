#!/usr/bin/env python3
"""Interleaved A/B timing of kernel variants in ONE process (cdna guide rule 24).
usage: python tools/ab_variants.py VAR=val1,val2,... [--shape B,H,N,D] [--rounds R] [--iters I]
Each variant = one value of the environment variable VAR (read by the library per launch)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from attention_mechanisms.fastmax import fastmax
from fastmax_experiments_amd import _lib

TUNE_KEYS = {"FASTMAX_MFMA_VARIANT": b"mfma_variant"}


def set_variant(var, x):
    """library knobs are read from the environment once; afterwards they change through fastmax_hip_tune"""
    if var in TUNE_KEYS:
        _lib.check(_lib.lib().fastmax_hip_tune(TUNE_KEYS[var], int(x)), "fastmax_hip_tune")
    else:
        os.environ[var] = x


def main():
    var, vals = sys.argv[1].split("=")
    vals = vals.split(",")
    shape = (16, 32, 4096, 64)
    rounds, iters = 7, 20
    args = sys.argv[2:]
    for i, a in enumerate(args):
        if a == "--shape":
            shape = tuple(int(x) for x in args[i + 1].split(","))
        if a == "--rounds":
            rounds = int(args[i + 1])
        if a == "--iters":
            iters = int(args[i + 1])
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v = (torch.randn(*shape, device="cuda", generator=g) for _ in range(3))
    ref = None
    res = {x: [] for x in vals}
    for rd in range(rounds + 1):
        for x in vals:
            set_variant(var, x)
            o = fastmax(q, k, v)
            if ref is None:
                ref = o.clone()
            elif rd == 0:
                print(f"{var}={x}: max |diff| vs first variant = {float((o - ref).abs().max()):.3e}")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fastmax(q, k, v)
            e1.record()
            torch.cuda.synchronize()
            if rd > 0:
                res[x].append(e0.elapsed_time(e1) / iters * 1e3)
    B, H, N, D = shape
    byts = 16 * B * H * N * D
    print(f"shape {shape}, {rounds} rounds x {iters} iters, algorithmic bytes {byts}")
    for x in vals:
        med, mn = statistics.median(res[x]), min(res[x])
        print(f"{var}={x}: median {med:.1f} us  min {mn:.1f} us   -> {byts / med / 1e3:.0f} GB/s median, "
              f"{byts / mn / 1e3:.0f} GB/s best  ({byts / med / 1e3 / 8000 * 100:.1f}% of 8 TB/s)")


if __name__ == "__main__":
    main()
