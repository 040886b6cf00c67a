#!/usr/bin/env python3
"""Per-launch durations of the headline forward from an IDLE device (what `bench.py --steps 20 --warmup 5` sees): the chip
starts at boost clocks, the power controller then pulls them down and settles, so the first ~25 launches are a transient.
usage: python tools/transient.py VAR=val1,val2,... [--launches 40] [--idle 2.0] [--repeat 2] [--shape B,H,N,D]
Prints one line of per-launch microseconds per (variant, repeat) and the mean over launches 6..25 (the bench's timed region)."""
import os
import sys
import time

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
    shape, launches, idle, repeat = (16, 32, 4096, 64), 40, 2.0, 2
    args = sys.argv[2:]
    for i, a in enumerate(args):
        if a == "--shape":
            shape = tuple(int(x) for x in args[i + 1].split(","))
        if a == "--launches":
            launches = int(args[i + 1])
        if a == "--idle":
            idle = float(args[i + 1])
        if a == "--repeat":
            repeat = int(args[i + 1])
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v = (torch.randn(*shape, device="cuda", generator=g) for _ in range(3))
    for x in vals:                                   # compile / attribute-set / allocator warm: one launch each, then idle
        set_variant(var, x)
        fastmax(q, k, v)
    torch.cuda.synchronize()
    for rp in range(repeat):
        for x in vals:
            set_variant(var, x)
            time.sleep(idle)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(launches + 1)]
            ev[0].record()
            for i in range(launches):
                fastmax(q, k, v)
                ev[i + 1].record()
            torch.cuda.synchronize()
            us = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(launches)]
            timed = us[5:25]
            print(f"{var}={x} rep{rp}: bench-window mean {sum(timed) / len(timed):.1f} us, last-10 mean "
                  f"{sum(us[-10:]) / 10:.1f} us | " + " ".join(f"{u:.0f}" for u in us), flush=True)


if __name__ == "__main__":
    main()
