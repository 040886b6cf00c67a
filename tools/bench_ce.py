#!/usr/bin/env python3
"""Time the cross-entropy kernels against torch's on the fine-tune shapes (tokens x vocabulary logits)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from fastmax_experiments_amd.loss import chunked_cross_entropy


def timeit(fn, iters=5, rounds=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters)
    return statistics.median(ts)


print("| tokens | vocab | dtype | HIP fwd+bwd ms | GB/s (3 passes) | torch cross_entropy fwd+bwd ms |")
print("|---|---|---|---|---|---|")
for M, V, dt in ((4096, 32000, torch.bfloat16), (16384, 32000, torch.bfloat16), (4096, 32000, torch.float32), (8192, 50304, torch.bfloat16)):
    logits = torch.randn(1, M, V, device="cuda").to(dt).requires_grad_(True)
    targets = torch.randint(0, V, (1, M), device="cuda")

    def ours():
        logits.grad = None
        chunked_cross_entropy(logits, targets).backward()

    def theirs():
        logits.grad = None
        F.cross_entropy(logits.reshape(-1, V), targets.reshape(-1), ignore_index=-1).backward()
    a, b = timeit(ours), timeit(theirs)
    byts = 3 * M * V * logits.element_size()
    print(f"| {M} | {V} | {str(dt).split('.')[-1]} | {a:.3f} | {byts / a / 1e6:.0f} | {b:.3f} |", flush=True)
