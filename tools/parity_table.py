#!/usr/bin/env python3
"""Measured forward / backward errors of every kernel family against the C oracle (fp64 accumulate) on the same rounded
inputs, next to the error that rounding the exact result to the storage dtype alone would give.  Markdown to stdout."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from attention_mechanisms.fastmax import fastmax
from attention_mechanisms.fastmax_hack import fastmax_hack
from fastmax_experiments_amd import _lib, ops
from oracle import c_oracle, fastmax_oracle as orc

DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}
PATH = {"auto": _lib.PATH_AUTO, "tiles": _lib.PATH_QUADRATIC_MFMA, "valu": _lib.PATH_QUADRATIC, "recurrent": _lib.PATH_RECURRENT}


def nw(a, r):
    return np.linalg.norm(a - r) / max(np.linalg.norm(r), 1e-30)


cases = [
    # label, op, path, (B,H,N,D), dtype, p, mask
    ("p=1 linear scan, headline kernel", "fastmax", "auto", (1, 4, 4096, 64), "f32", 1, True),
    ("p=1 linear scan", "fastmax", "auto", (1, 4, 4096, 64), "bf16", 1, True),
    ("p=1 linear scan", "fastmax", "auto", (1, 4, 4096, 64), "f16", 1, True),
    ("p=1 linear scan, D=128 (8 waves)", "fastmax", "auto", (1, 4, 4096, 128), "bf16", 1, True),
    ("p=1 D=128 fp32: two-part forward, role-swapped scan backward (round 2)", "fastmax", "auto", (1, 2, 4096, 128), "f32", 1, True),
    ("p=1 D=128 fp16: same", "fastmax", "auto", (1, 2, 4096, 128), "f16", 1, True),
    ("p=1 D=96 fp32: same, padded", "fastmax", "auto", (1, 2, 2048, 96), "f32", 1, True),
    ("p=1 linear scan, padded head size", "fastmax", "auto", (1, 4, 2048, 48), "f32", 1, True),
    ("p=1 sequence split (4 heads x 16k)", "fastmax", "auto", (1, 4, 16384, 64), "f32", 1, True),
    ("linearmax (fused prologue fwd, HIP prologue bwd)", "linearmax", "auto", (1, 4, 4096, 64), "bf16", 1, True),
    ("linearmax", "linearmax", "auto", (1, 4, 4096, 64), "f32", 1, True),
    ("p=2 tiles 32x32x16", "fastmax", "auto", (1, 4, 2048, 64), "f32", 2, True),
    ("p=2 tiles 32x32x16", "fastmax", "auto", (1, 4, 2048, 64), "bf16", 2, True),
    ("p=2 tiles 32x32x16", "fastmax", "auto", (1, 4, 2048, 64), "f16", 2, True),
    ("p=2 tiles 32x32x16, D=128", "fastmax", "auto", (1, 2, 2048, 128), "bf16", 2, True),
    ("p=2 unmasked", "fastmax", "auto", (1, 4, 1024, 64), "f32", 2, False),
    ("p=1 unmasked", "fastmax", "auto", (1, 4, 1024, 64), "bf16", 1, False),
    ("p=2 tiles 16x16x32 (N < 256)", "fastmax", "auto", (2, 4, 200, 64), "f32", 2, True),
    ("p=1 forced tile kernels", "fastmax", "tiles", (1, 4, 2048, 64), "f32", 1, True),
    ("vector-ALU family", "fastmax", "valu", (1, 2, 300, 64), "f32", 2, True),
    ("vector-ALU recurrent", "fastmax", "recurrent", (1, 2, 1000, 64), "f32", 1, True),
]
print("| kernel family | (B,H,N,D) | dtype | p | mask | fwd error | fwd rounding floor | dQ | dK | dV | grad rounding floor |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for label, op, path, shape, dt, p, mask in cases:
    ops.set_forced_path(PATH[path])
    tdt = DT[dt]
    g = torch.Generator().manual_seed(shape[2])
    q, k, v, go = (torch.randn(shape, generator=g).to(tdt) for _ in range(4))
    qq, kk, vv = (t.cuda().requires_grad_(True) for t in (q, k, v))
    f = fastmax_hack if op == "linearmax" else fastmax
    o = f(qq, kk, vv, p=p, mask=mask)
    o.backward(go.cuda().to(o.dtype))
    qn, kn, vn, gn = (t.float().numpy() for t in (q, k, v, go))
    if op == "linearmax":
        ro = orc.linearmax_fwd(qn, kn, vn, chunk=64)
        grads = None
    else:
        ro, _ = c_oracle.fwd(qn, kn, vn, mask=mask, p=p)
        grads = c_oracle.bwd(qn, kn, vn, gn, mask=mask, p=p)
    ro = np.asarray(ro, dtype=np.float64)
    odt = o.dtype
    floor = nw(torch.from_numpy(ro).to(odt).double().numpy(), ro)
    row = f"| {label} | {shape} | {dt} | {p} | {mask} | {nw(o.detach().double().cpu().numpy(), ro):.2e} | {floor:.2e} |"
    if grads is not None:
        errs = [nw(t.grad.double().cpu().numpy(), np.asarray(r, dtype=np.float64)) for t, r in zip((qq, kk, vv), grads)]
        gfloor = nw(torch.from_numpy(np.asarray(grads[0], dtype=np.float64)).to(tdt).double().numpy(), np.asarray(grads[0], dtype=np.float64))
        row += f" {errs[0]:.2e} | {errs[1]:.2e} | {errs[2]:.2e} | {gfloor:.2e} |"
    else:
        row += " (autograd test) | | | |"
    print(row, flush=True)
ops.set_forced_path(_lib.PATH_AUTO)
