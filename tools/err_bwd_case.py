#!/usr/bin/env python3
"""Backward error of one case against the C oracle (fp64) on the same rounded inputs, next to the error that rounding
the exact gradients to the storage dtype alone would give.  usage: err_bwd_case.py op B H N D dtype p"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from attention_mechanisms.fastmax import fastmax
from oracle import c_oracle
op, B, H, N, D, dt, p = sys.argv[1], *map(int, sys.argv[2:6]), sys.argv[6], int(sys.argv[7])
tdt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[dt]
for seed, scale in ((0, 1.0), (1, 3.0)):
    g = torch.Generator().manual_seed(seed)
    q, k, v, go = ((torch.randn(B, H, N, D, generator=g) * scale).to(tdt) for _ in range(4))
    qq, kk, vv = (t.cuda().requires_grad_(True) for t in (q, k, v))
    o = fastmax(qq, kk, vv, mask=True, p=p)
    o.backward(go.cuda())
    e = c_oracle.bwd(q.float().numpy(), k.float().numpy(), v.float().numpy(), go.float().numpy(), mask=True, p=p)
    for t, r, n in zip((qq, kk, vv), e, ("dq", "dk", "dv")):
        got = t.grad.float().cpu().numpy()
        rnd = torch.from_numpy(r).to(tdt).float().numpy()
        print(f"seed {seed} {n}: normwise {np.linalg.norm(got - r) / np.linalg.norm(r):.3e}   rounding alone {np.linalg.norm(rnd - r) / np.linalg.norm(r):.3e}", flush=True)
