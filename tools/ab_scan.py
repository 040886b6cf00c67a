#!/usr/bin/env python3
"""Time the linear-time (p=1 masked) scan cases: forward and forward+backward, fastmax and linearmax.
    python tools/ab_scan.py [substring]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from attention_mechanisms.fastmax import fastmax
from attention_mechanisms.fastmax_hack import fastmax_hack

DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}
CASES = [
    ("C5 linearmax 16k", "linearmax", (1, 32, 16384, 128), "bf16"),
    ("C5 fastmax p=1 16k", "fastmax", (1, 32, 16384, 128), "bf16"),
    ("C4 linearmax", "linearmax", (2, 32, 4096, 128), "bf16"),
    ("C3 linearmax", "linearmax", (8, 32, 2048, 64), "bf16"),
    ("headline bf16 p=1", "fastmax", (16, 32, 4096, 64), "bf16"),
    ("headline f32 p=1", "fastmax", (16, 32, 4096, 64), "f32"),
    ("D=128 many heads bf16", "fastmax", (16, 32, 4096, 128), "bf16"),
    ("C4 heads f32 p=1", "fastmax", (2, 32, 4096, 128), "f32"),
    ("C5 heads f32 p=1 16k", "fastmax", (1, 32, 16384, 128), "f32"),
    ("C5 heads f32 linearmax 16k", "linearmax", (1, 32, 16384, 128), "f32"),
]


def timeit(fn, iters=10, rounds=7):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters)
    return statistics.median(ts), min(ts)


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    print("| case | shape | dtype | fwd ms (median/min) | fwd GB/s (4 passes) | fwd+bwd ms (median/min) |")
    print("|---|---|---|---|---|---|")
    for name, op, shape, dt in CASES:
        if only and only not in name:
            continue
        g = torch.Generator(device="cuda").manual_seed(0)
        q, k, v, go = (torch.randn(*shape, device="cuda", generator=g).to(DT[dt]) for _ in range(4))
        f = fastmax_hack if op == "linearmax" else fastmax

        def fwd():
            with torch.no_grad():
                f(q, k, v, p=1, mask=True)

        qg, kg, vg = (t.clone().requires_grad_(True) for t in (q, k, v))

        def both():
            o = f(qg, kg, vg, p=1, mask=True)
            qg.grad = kg.grad = vg.grad = None
            o.backward(go)

        a, b = timeit(fwd), timeit(both)
        byts = 4 * q.numel() * q.element_size()
        print(f"| {name} | {shape} | {dt} | {a[0]:.3f} / {a[1]:.3f} | {byts / a[0] / 1e6:.0f} | {b[0]:.3f} / {b[1]:.3f} |", flush=True)
        del q, k, v, go, qg, kg, vg
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
