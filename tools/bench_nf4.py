#!/usr/bin/env python3
"""Time the fused NF4 + LoRA linear (forward and dx) against a dense bf16 library GEMM of the same shape."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fastmax_experiments_amd import lora


def timeit(fn, iters=10, rounds=5):
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


def main():
    print("| M (tokens) | K | N | NF4+LoRA fwd ms | TFLOP/s | dx ms | TFLOP/s | dense bf16 F.linear ms | TFLOP/s |")
    print("|---|---|---|---|---|---|---|---|---|")
    for M, K, N in ((64, 4096, 4096), (256, 4096, 4096), (512, 4096, 4096), (1024, 4096, 4096), (2048, 4096, 4096), (4096, 4096, 4096), (2048, 4096, 11008), (16384, 2048, 2560), (16384, 4096, 4096), (8192, 4096, 12288), (16384, 4096, 11008)):
        torch.manual_seed(0)
        layer = lora.LoRALinear(K, N, r=8, lora_alpha=16, bias=False)
        torch.nn.init.normal_(layer.lora_B, std=0.02)
        dense_w = layer.linear.weight.data.to("cuda", torch.bfloat16)
        layer.quantize_base().cuda()
        if os.environ.get("FASTMAX_NF4_CACHE") == "1":
            layer.linear.cache_dense()
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        gy = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
        with torch.no_grad():
            t_f = timeit(lambda: layer(x))
            t_d = timeit(lambda: torch.nn.functional.linear(x, dense_w))
        wq, am = layer.linear.weight.data, layer.linear.weight.quant_state[0]
        fn = lora._QLoRALinearFn
        wd = layer.linear._dense_cache

        def dx():
            xx = x.detach().requires_grad_(True)
            y = fn.apply(xx, None, None, wq, am, None, N, K, wd)
            y.backward(gy)
        t_b = timeit(dx) - timeit(lambda: fn.apply(x, None, None, wq, am, None, N, K, wd))
        fl = 2 * M * K * N / 1e9
        print(f"| {M} | {K} | {N} | {t_f:.3f} | {fl / t_f:.0f} | {t_b:.3f} | {fl / t_b:.0f} | {t_d:.3f} | {fl / t_d:.0f} |", flush=True)


if __name__ == "__main__":
    main()
