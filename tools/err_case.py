#!/usr/bin/env python3
"""Forward error of one case against the C oracle (fp64 accumulate) on the same rounded inputs.
usage: err_case.py op B H N D dtype p   -> prints normwise and max-abs-relative-to-scale errors"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from attention_mechanisms.fastmax import fastmax
from oracle import c_oracle
op, B, H, N, D, dt, p = sys.argv[1], *map(int, sys.argv[2:6]), sys.argv[6], int(sys.argv[7])
tdt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[dt]
for seed, scale in ((0, 1.0), (1, 3.0), (2, 0.3)):
    g = torch.Generator().manual_seed(seed)
    q, k, v = ((torch.randn(B, H, N, D, generator=g) * scale + (0.5 if seed == 1 else 0.0)).to(tdt) for _ in range(3))
    o = fastmax(q.cuda(), k.cuda(), v.cuda(), mask=True, p=p).float().cpu().numpy()
    ro, _ = c_oracle.fwd(q.float().numpy(), k.float().numpy(), v.float().numpy(), mask=True, p=p)
    err = o - ro
    print(f"seed {seed} scale {scale}: normwise {np.linalg.norm(err) / np.linalg.norm(ro):.3e}   max|err|/max|ref| {np.abs(err).max() / np.abs(ro).max():.3e}   "
          f"bf16-rounding-of-ref normwise {np.linalg.norm(torch.from_numpy(ro).to(tdt).float().numpy() - ro) / np.linalg.norm(ro):.3e}", flush=True)
