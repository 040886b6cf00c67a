#!/usr/bin/env python3
"""normalize_backward under the default (row pass + fix-up) and FASTMAX_NORMALIZE_BWD_TWO_PASS=1 (reduce + apply) forms:
run once per setting with an output file, then with both files to compare.  usage: check_normalize_bwd.py out.pt | a.pt b.pt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if len(sys.argv) == 3:
    a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
    worst = 0.0
    for x, y in zip(a, b):
        d = float((x.float() - y.float()).abs().max() / y.float().abs().max())
        frac = float((x != y).float().mean())
        print(f"{tuple(x.shape)} {x.dtype}: max |diff| / max |value| = {d:.2e}, {100 * frac:.3f} % of elements differ")
        worst = max(worst, d)
    sys.exit(1 if worst > 1e-2 else 0)
from fastmax_experiments_amd import ops
outs = []
for shape, dt in [((2, 3, 700, 64), torch.bfloat16), ((1, 2, 256, 32), torch.float32), ((1, 1, 1030, 128), torch.float16),
                  ((8, 32, 2048, 64), torch.bfloat16), ((1, 2, 5, 64), torch.float32)]:
    torch.manual_seed(1)
    x = (torch.randn(shape, device="cuda") * 2 + 0.3).to(dt)
    gy = torch.randn(shape, device="cuda").to(dt)
    y, inv = ops.normalize_cast(x)
    outs.append(ops.normalize_backward(x, gy, inv).cpu())
torch.save(outs, sys.argv[1])
