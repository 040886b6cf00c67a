#!/usr/bin/env python3
"""The rank-r kernels of csrc/lora_thin.hip against the tensor ops they replace: values and time.
usage: bench_lora_thin.py [M] [K] [N] [R]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastmax_experiments_amd import lora

M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2560
R = int(sys.argv[4]) if len(sys.argv) > 4 else 16
RP = 16 if R <= 16 else 32
torch.manual_seed(0)
dev = "cuda"
x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
A = torch.zeros(RP, K, device=dev, dtype=torch.bfloat16)
A[:R] = torch.randn(R, K, device=dev) * 0.05
eb = torch.zeros(N, RP, device=dev, dtype=torch.bfloat16)
eb[:, :R] = torch.randn(N, R, device=dev) * 0.05


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def rel(a, b):
    return float((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-30))


print(f"M={M} K={K} N={N} R={R} (RP={RP})")
# down
ea, eat = lora.lora_down(x, A)
ref = (x.float() @ A.float().t())
print(f"down  e   rel {rel(ea, ref):.2e}   e^T rel {rel(eat[:, :M], ref.t()):.2e}  pad zero {bool((eat[:, M:] == 0).all())}")
t1 = timeit(lambda: lora.lora_down(x, A))
t0 = timeit(lambda: x @ A.t())
print(f"down  x A^T       : hip {t1:7.1f} us   torch {t0:7.1f} us   ({M * K * 2 / t1 / 1e6:.2f} TB/s)")
ebt = eb.t().contiguous()
d_ea, d_eat = lora.lora_down(dy, ebt)
ref2 = dy.float() @ eb.float()
print(f"down  dy eb rel {rel(d_ea, ref2):.2e}")
t1 = timeit(lambda: lora.lora_down(dy, ebt))
t0 = timeit(lambda: dy @ eb)
print(f"down  dy eb       : hip {t1:7.1f} us   torch {t0:7.1f} us   ({M * N * 2 / t1 / 1e6:.2f} TB/s)")
# tn
c = lora.lora_tn(eat, dy)
ct = lora.lora_tn(eat, dy, R, torch.bfloat16, transpose=True)
refc = ea.float().t() @ dy.float()
print(f"tn    ea^T dy rel {rel(c, refc):.2e}   transposed bf16 [:R] rel {rel(ct, refc[:R].t()):.2e}")
t1 = timeit(lambda: lora.lora_tn(eat, dy))
t0 = timeit(lambda: dy.t() @ ea)
print(f"tn    ea^T dy     : hip {t1:7.1f} us   torch {t0:7.1f} us   ({M * N * 2 / t1 / 1e6:.2f} TB/s)")
c2 = lora.lora_tn(d_eat, x)
refc2 = d_ea.float().t() @ x.float()
print(f"tn    d_ea^T x rel {rel(c2, refc2):.2e}")
t1 = timeit(lambda: lora.lora_tn(d_eat, x))
t0 = timeit(lambda: d_ea.t() @ x)
print(f"tn    d_ea^T x    : hip {t1:7.1f} us   torch {t0:7.1f} us   ({M * K * 2 / t1 / 1e6:.2f} TB/s)")
# up
bias = torch.randn(N, device=dev)
y0 = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
y = y0.clone()
lora.lora_up_(y, ea, eb, bias)
refy = y0.float() + ea.float() @ eb.float().t() + bias
print(f"up    y += ea eb^T + bias rel {rel(y, refy):.2e}")
y = y0.clone()
t1 = timeit(lambda: lora.lora_up_(y, ea, eb))
y = y0.clone()
t0 = timeit(lambda: y.addmm_(ea, eb.t()))
print(f"up    y += ea eb^T: hip {t1:7.1f} us   torch {t0:7.1f} us   ({M * N * 4 / t1 / 1e6:.2f} TB/s)")
at = A.t().contiguous()
dx0 = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
dx = dx0.clone()
lora.lora_up_(dx, d_ea, at)
refdx = dx0.float() + d_ea.float() @ A.float()
print(f"up    dx += d_ea A rel {rel(dx, refdx):.2e}")
t1 = timeit(lambda: lora.lora_up_(dx, d_ea, at))
t0 = timeit(lambda: dx.addmm_(d_ea, A))
print(f"up    dx += d_ea A: hip {t1:7.1f} us   torch {t0:7.1f} us   ({M * K * 4 / t1 / 1e6:.2f} TB/s)")
# ragged rows
for m in (1, 17, 130, 2049):
    xs, dys = x[:m], dy[:m]
    e1, et1 = lora.lora_down(xs, A)
    r1 = rel(e1, xs.float() @ A.float().t())
    c1 = lora.lora_tn(et1, dys)
    r2 = rel(c1, e1.float().t() @ dys.float())
    ys = y0[:m].clone()
    lora.lora_up_(ys, e1, eb)
    r3 = rel(ys, y0[:m].float() + e1.float() @ eb.float().t())
    print(f"rows {m:5d}: down {r1:.2e}  tn {r2:.2e}  up {r3:.2e}")
