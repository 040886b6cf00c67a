#!/usr/bin/env python3
"""Run one operator case a few times (for rocprofv3 --kernel-trace --stats). usage: trace_case.py op B H N D dtype p"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from attention_mechanisms.fastmax import fastmax
from attention_mechanisms.fastmax_hack import fastmax_hack
op, B, H, N, D, dt, p = sys.argv[1], *map(int, sys.argv[2:6]), sys.argv[6], int(sys.argv[7])
tdt = {"f32": torch.float32, "bf16": torch.bfloat16}[dt]
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = (torch.randn(B, H, N, D, device="cuda", generator=g).to(tdt) for _ in range(3))
with torch.no_grad():
    for _ in range(12):
        o = fastmax_hack(q, k, v, p=p, mask=True) if op == "linearmax" else fastmax(q, k, v, p=p, mask=True)
torch.cuda.synchronize()
