#!/usr/bin/env python3
"""Time the p=2 (tile-kernel) cases: forward and forward+backward.  One process per library build:
    python tools/ab_quad32.py                       # the tree's library
    FASTMAX_LIB_PATH=tools/_ab/libfastmax_hip_r02.so python tools/ab_quad32.py
Prints one markdown row per case."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from attention_mechanisms.fastmax import fastmax

DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}
CASES = [
    ("headline p=2", (16, 32, 4096, 64), "bf16", 2, True),
    ("C3 heads p=2", (8, 32, 2048, 64), "bf16", 2, True),
    ("C4 heads p=2", (2, 32, 4096, 128), "bf16", 2, True),
    ("C2 heads p=2", (16, 4, 1024, 32), "bf16", 2, True),
    ("headline p=2 f32", (16, 32, 4096, 64), "f32", 2, True),
    ("headline p=1 tiles unmasked", (16, 32, 4096, 64), "bf16", 1, False),
    ("D=128 norm_term=4 (pow2 a? no)", (2, 32, 4096, 128), "bf16", 2, True),
]


def timeit(fn, iters, rounds=7):
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
    tag = os.environ.get("FASTMAX_LIB_PATH", "tree")
    only = sys.argv[1:] and sys.argv[1]
    print(f"| lib | case | shape | dtype | p | causal | fwd ms (median/min) | fwd+bwd ms (median/min) |")
    print("|---|---|---|---|---|---|---|---|")
    for name, shape, dt, p, causal in CASES:
        if only and only not in name:
            continue
        g = torch.Generator(device="cuda").manual_seed(0)
        q, k, v, go = (torch.randn(*shape, device="cuda", generator=g).to(DT[dt]) for _ in range(4))

        def fwd():
            with torch.no_grad():
                fastmax(q, k, v, mask=causal, p=p)

        qg, kg, vg = (t.clone().requires_grad_(True) for t in (q, k, v))

        def both():
            o = fastmax(qg, kg, vg, mask=causal, p=p)
            qg.grad = kg.grad = vg.grad = None
            o.backward(go.to(o.dtype))

        iters = 5 if dt == "f32" else 10
        f = timeit(fwd, iters)
        b = timeit(both, iters)
        print(f"| {os.path.basename(tag)} | {name} | {shape} | {dt} | {p} | {causal} | {f[0]:.3f} / {f[1]:.3f} | {b[0]:.3f} / {b[1]:.3f} |", flush=True)
        del q, k, v, go, qg, kg, vg
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
