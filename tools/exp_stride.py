#!/usr/bin/env python3
"""Does the head stride matter (HBM channel alignment of the 512 concurrent streams)?  Time the headline forward with
q, k, v views into tensors padded along N.  usage: python tools/exp_stride.py"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastmax_experiments_amd import ops

B, H, N, D = 16, 32, 4096, 64
for pad in (0, 0, 16, 48, 80, 4, 0):
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v = (torch.randn(B, H, N + pad, D, device="cuda", generator=g)[:, :, :N] for _ in range(3))
    for _ in range(5):
        ops.forward(q, k, v, 1, True, 8.0 * D ** 0.5, float(N), torch.float32, need_g=False)
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.forward(q, k, v, 1, True, 8.0 * D ** 0.5, float(N), torch.float32, need_g=False)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20)
    print(f"pad {pad:3d} rows (head stride {(N + pad) * D * 4} B): {statistics.median(ts) * 1e3:7.1f} us  (min {min(ts) * 1e3:.1f})", flush=True)
