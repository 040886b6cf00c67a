#!/usr/bin/env python3
"""Time the operator at the BASELINE.json config shapes (parity-test cases, not the headline bench line).
Writes a markdown table to stdout.  usage: python tools/bench_shapes.py [--quick]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from attention_mechanisms.fastmax import fastmax
from attention_mechanisms.fastmax_hack import fastmax_hack
from fastmax_experiments_amd import _lib, ops

DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}


def timeit(fn, iters, rounds=5):
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
    return statistics.median(ts)


def main():
    quick = "--quick" in sys.argv
    cases = [
        # name, op, (B,H,N,D), dtype, p, mode
        ("headline fastmax p=1", "fastmax", (16, 32, 4096, 64), "f32", 1, "fwd"),
        ("headline B=4", "fastmax", (4, 32, 4096, 64), "f32", 1, "fwd"),
        ("headline B=1", "fastmax", (1, 32, 4096, 64), "f32", 1, "fwd"),
        ("headline bf16", "fastmax", (16, 32, 4096, 64), "bf16", 1, "fwd"),
        ("headline p=2", "fastmax", (16, 32, 4096, 64), "f32", 2, "fwd"),
        ("headline p=2 bf16", "fastmax", (16, 32, 4096, 64), "bf16", 2, "fwd"),
        ("headline p=1 fwd+bwd", "fastmax", (16, 32, 4096, 64), "f32", 1, "fwd+bwd"),
        ("headline p=2 fwd+bwd bf16", "fastmax", (16, 32, 4096, 64), "bf16", 2, "fwd+bwd"),
        ("C2 pythia-14m heads fastmax(p=2)", "fastmax", (16, 4, 1024, 32), "bf16", 2, "fwd"),
        ("C2 pythia-14m heads linearmax", "linearmax", (16, 4, 1024, 32), "bf16", 1, "fwd"),
        ("C2 linearmax, HIP graph replay", "linearmax_graph", (16, 4, 1024, 32), "bf16", 1, "fwd"),
        ("C3 tinyllama heads fastmax(p=2)", "fastmax", (8, 32, 2048, 64), "bf16", 2, "fwd"),
        ("C3 tinyllama heads fastmax(p=2) fwd+bwd", "fastmax", (8, 32, 2048, 64), "bf16", 2, "fwd+bwd"),
        ("C3 tinyllama heads linearmax", "linearmax", (8, 32, 2048, 64), "bf16", 1, "fwd"),
        ("C4 llama-2-7b heads fastmax(p=2)", "fastmax", (2, 32, 4096, 128), "bf16", 2, "fwd"),
        ("C4 llama-2-7b heads fastmax(p=2) fwd+bwd", "fastmax", (2, 32, 4096, 128), "bf16", 2, "fwd+bwd"),
        ("C5 llama-2-7b linearmax 16k", "linearmax", (1, 32, 16384, 128), "bf16", 1, "fwd"),
        ("C5 fastmax p=1 16k (no prologue)", "fastmax", (1, 32, 16384, 128), "bf16", 1, "fwd"),
        ("C3 linearmax fwd+bwd (training route)", "linearmax", (8, 32, 2048, 64), "bf16", 1, "fwd+bwd"),
        ("C4 linearmax fwd+bwd", "linearmax", (2, 32, 4096, 128), "bf16", 1, "fwd+bwd"),
        ("C5 linearmax 16k fwd+bwd", "linearmax", (1, 32, 16384, 128), "bf16", 1, "fwd+bwd"),
        ("headline bf16 p=1 fwd+bwd", "fastmax", (16, 32, 4096, 64), "bf16", 1, "fwd+bwd"),
        ("headline p=2 fwd+bwd f32", "fastmax", (16, 32, 4096, 64), "f32", 2, "fwd+bwd"),
        ("C4 heads fastmax p=1 f32 fwd+bwd (scan backward)", "fastmax", (2, 32, 4096, 128), "f32", 1, "fwd+bwd"),
        ("C5 heads fastmax p=1 f32 16k fwd+bwd (scan backward)", "fastmax", (1, 32, 16384, 128), "f32", 1, "fwd+bwd"),
        ("C5 heads linearmax f32 16k fwd+bwd", "linearmax", (1, 32, 16384, 128), "f32", 1, "fwd+bwd"),
        ("pythia-1b heads (head size 256) fastmax(p=2)", "fastmax", (8, 8, 2048, 256), "bf16", 2, "fwd"),
        ("pythia-1b heads (head size 256) fastmax(p=2) fwd+bwd", "fastmax", (8, 8, 2048, 256), "bf16", 2, "fwd+bwd"),
        ("pythia-1b heads linearmax fwd+bwd", "linearmax", (8, 8, 2048, 256), "bf16", 1, "fwd+bwd"),
    ]
    if quick:
        cases = cases[:4]
    print("| case | (B,H,N,D) | dtype | p | mode | kernel path | ms | M tokens/s | algorithmic GB/s | % of 8 TB/s |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for name, op, shape, dt, p, mode in cases:
        B, H, N, D = shape
        g = torch.Generator(device="cuda").manual_seed(0)
        q, k, v, go = (torch.randn(*shape, device="cuda", generator=g).to(DT[dt]) for _ in range(4))
        train = mode == "fwd+bwd"
        if train:
            q.requires_grad_(True), k.requires_grad_(True), v.requires_grad_(True)

        graph = None
        if op == "linearmax_graph":                      # launch-bound shape: record the five launches once, replay
            with torch.no_grad():
                fastmax_hack(q, k, v, p=p, mask=True)
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    fastmax_hack(q, k, v, p=p, mask=True)
                torch.cuda.current_stream().wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    fastmax_hack(q, k, v, p=p, mask=True)

        def step():
            if graph is not None:
                graph.replay()
                return
            with torch.set_grad_enabled(train):
                o = fastmax_hack(q, k, v, p=p, mask=True) if op == "linearmax" else fastmax(q, k, v, mask=True, p=p)
                if train:
                    q.grad = k.grad = v.grad = None
                    o.backward(go)

        path = _lib.PATH_NAMES.get(ops.selected_path(q, k, p, True), "?")
        if op.startswith("linearmax"):
            path = "mfma+fused prologue" if not train else path
        ms = timeit(step, 10 if ms_guess(shape, p) < 20 else 3)
        es = 4 if dt == "f32" else 2
        byts = (4 if not train else 12) * B * H * N * D * es
        print(f"| {name} | {shape} | {dt} | {p} | {mode} | {path} | {ms:.3f} | {B * N / ms / 1e3:.1f} | "
              f"{byts / ms / 1e6:.0f} | {byts / ms / 1e6 / 8000 * 100:.1f} |", flush=True)
        del q, k, v, go
        torch.cuda.empty_cache()


def ms_guess(shape, p):
    B, H, N, D = shape
    return B * H * N * (N if p == 2 else 64) * D / 2e12


if __name__ == "__main__":
    main()
