#!/usr/bin/env python3
"""Generation-time shapes: N_q new tokens against N_k cached keys (unmasked, lit_gpt/model.py:464-466), and the opt-in
decode state cache.  Markdown to stdout."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from attention_mechanisms.fastmax import fastmax
from fastmax_experiments_amd.decode import FastmaxDecodeState


def timeit(fn, iters=20, rounds=5):
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


print("| case | (B,H,Nq,Nk,D) | dtype | p | ms |")
print("|---|---|---|---|---|")
for B, H, Nq, Nk, D, dt, p in ((1, 32, 1, 4096, 64, torch.bfloat16, 2), (1, 32, 1, 4096, 128, torch.bfloat16, 2), (8, 32, 1, 4096, 64, torch.bfloat16, 2),
                               (1, 32, 1, 16384, 128, torch.bfloat16, 1), (1, 32, 16, 4096, 64, torch.bfloat16, 2), (1, 32, 128, 4096, 64, torch.bfloat16, 2)):
    q = torch.randn(B, H, Nq, D, device="cuda").to(dt)
    k, v = (torch.randn(B, H, Nk, D, device="cuda").to(dt) for _ in range(2))
    with torch.no_grad():
        ms = timeit(lambda: fastmax(q, k, v, mask=False, p=p))
    print(f"| unmasked over the KV cache | ({B},{H},{Nq},{Nk},{D}) | {str(dt).split('.')[-1]} | {p} | {ms:.4f} |", flush=True)
for B, H, T, D in ((1, 32, 4096, 64), (1, 32, 16384, 128), (8, 32, 4096, 64)):
    q, k, v = (torch.randn(B, H, T, D, device="cuda").to(torch.bfloat16) for _ in range(3))
    st = FastmaxDecodeState(B, H, D, device="cuda")
    st.prefill(q, k, v)
    q1, k1, v1 = (torch.randn(B, H, 1, D, device="cuda").to(torch.bfloat16) for _ in range(3))
    with torch.no_grad():
        ms = timeit(lambda: st.step(q1, k1, v1))
    print(f"| decode state cache step (p=1, opt-in) | ({B},{H},1,{T},{D}) | bfloat16 | 1 | {ms:.4f} |", flush=True)

# the frozen 4-bit linear at generation-size row counts (merged weights: no LoRA branch), eager and as a HIP graph replay
from fastmax_experiments_amd import lora
for M, K, N in ((1, 4096, 4096), (16, 4096, 4096), (1, 4096, 11008), (1, 11008, 4096)):
    lin = torch.nn.Linear(K, N, bias=False)
    q4 = lora.NF4Linear.from_linear(lin).cuda()
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    wd = q4.dequantize(torch.bfloat16)
    with torch.no_grad():
        t_e = timeit(lambda: q4(x))
        t_d = timeit(lambda: torch.nn.functional.linear(x, wd))
        q4(x)
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            q4(x)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(10):
                y = q4(x)
        t_g = timeit(lambda: graph.replay()) / 10
    print(f"| NF4 linear (M,K,N)=({M},{K},{N}): eager {t_e * 1e3:.1f} us, graph replay {t_g * 1e3:.1f} us = {N * K / 2 / t_g / 1e6:.0f} GB/s of codes; bf16 F.linear {t_d * 1e3:.1f} us | | bfloat16 | | {t_g:.4f} |", flush=True)
