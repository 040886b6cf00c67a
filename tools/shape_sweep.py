#!/usr/bin/env python3
"""Outlier hunt: time the operator over an irregular set of shapes and print time per (head x token) so that a shape that falls
onto a slow path stands out."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from attention_mechanisms.fastmax import fastmax
from fastmax_experiments_amd import _lib, ops

DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}
cases = []
for dt in ("bf16", "f32"):
    for p in (1, 2):
        for mask in (True, False):
            for (B, H, Nq, Nk, D) in ((1, 1, 8192, 8192, 64), (64, 32, 256, 256, 64), (2, 32, 1000, 1000, 80), (4, 8, 3000, 3000, 40),
                                      (1, 32, 4096, 4096, 96), (16, 32, 100, 100, 64), (1, 32, 8, 2048, 64), (2, 16, 2048, 2048, 50),
                                      (1, 2, 65536, 65536, 64)):
                if mask and Nq != Nk:
                    continue
                if not mask and Nq * Nk > 1 << 27:
                    continue
                if p == 2 and Nq * Nk > 1 << 28:
                    continue
                cases.append((B, H, Nq, Nk, D, dt, p, mask))
print("| (B,H,Nq,Nk,D) | dtype | p | mask | path | fwd ms | ns per head-token | fwd+bwd ms |")
print("|---|---|---|---|---|---|---|---|")
for B, H, Nq, Nk, D, dt, p, mask in cases:
    tdt = DT[dt]
    q = torch.randn(B, H, Nq, D, device="cuda").to(tdt)
    k, v = (torch.randn(B, H, Nk, D, device="cuda").to(tdt) for _ in range(2))
    path = _lib.PATH_NAMES.get(ops.selected_path(q, k, p, mask), "?")

    def fwd():
        with torch.no_grad():
            fastmax(q, k, v, mask=mask, p=p)

    def train():
        qq, kk, vv = (t.detach().requires_grad_(True) for t in (q, k, v))
        o = fastmax(qq, kk, vv, mask=mask, p=p)
        o.backward(torch.ones_like(o))

    def timeit(fn):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 3
    tf, tt = timeit(fwd), timeit(train)
    print(f"| ({B},{H},{Nq},{Nk},{D}) | {dt} | {p} | {mask} | {path} | {tf:.3f} | {tf * 1e6 / (B * H * Nq):.1f} | {tt:.3f} |", flush=True)
