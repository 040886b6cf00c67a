"""lm-head + cross entropy at fine-tune sizes (library GEMM + HIP row kernel; logits kept for the backward pass or recomputed),
forward and forward+backward.  `python tools/bench_head.py [M K V]`"""
import sys
import time

import torch

sys.path.insert(0, ".")
from fastmax_experiments_amd.loss import _LMHeadLoss  # noqa: E402


def timeit(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


def main():
    shapes = [(16384, 2048, 32000), (8192, 4096, 32000)] if len(sys.argv) < 4 else [tuple(int(a) for a in sys.argv[1:4])]
    for M, K, V in shapes:
        g = torch.Generator(device="cuda").manual_seed(0)
        x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(V, K, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
        t = torch.randint(0, V, (M,), device="cuda", generator=g)
        for name, fn, keep in (("library + rows, recompute", _LMHeadLoss, False), ("library + rows, logits kept", _LMHeadLoss, True)):
            def fwd():
                with torch.no_grad():
                    return fn.apply(x, w, t, -1, 4096, False)

            def both():
                xa = x.detach().requires_grad_(True)
                fn.apply(xa, w, t, -1, 4096, keep).backward()

            print(f"M={M} K={K} V={V} {name:28s} fwd (no grad) {timeit(fwd):7.3f} ms   fwd+bwd {timeit(both):7.3f} ms   loss {float(fwd()):.4f}",
                  flush=True)


if __name__ == "__main__":
    main()
