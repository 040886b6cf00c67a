#!/usr/bin/env python3
"""Run one operator case a few times (for rocprofv3 --kernel-trace --stats).
usage: trace_case.py op B H N D dtype p [fwd|fwd+bwd]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from attention_mechanisms.fastmax import fastmax
from attention_mechanisms.fastmax_hack import fastmax_hack
op, B, H, N, D, dt, p = sys.argv[1], *map(int, sys.argv[2:6]), sys.argv[6], int(sys.argv[7])
mode = sys.argv[8] if len(sys.argv) > 8 else "fwd"
tdt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[dt]
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = (torch.randn(B, H, N, D, device="cuda", generator=g).to(tdt).requires_grad_(mode != "fwd") for _ in range(3))
f = fastmax_hack if op == "linearmax" else fastmax
for _ in range(12):
    if mode == "fwd":
        with torch.no_grad():
            o = f(q, k, v, p=p, mask=True)
    else:
        o = f(q, k, v, p=p, mask=True)
        o.backward(torch.ones_like(o))
        q.grad = k.grad = v.grad = None
torch.cuda.synchronize()
