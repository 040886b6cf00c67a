#!/usr/bin/env python3
"""Attention sub-layer (QLoRA qkv linear -> RoPE -> fastmax / linearmax -> proj), forward+backward, un-profiled (HIP events,
median of 7 rounds of 5), one child process per route:
    hand-written tile GEMM route (default)  vs  FASTMAX_QLORA_ROUTE=library (decode once + hipBLASLt + rank-r kernels)
at lora_dropout 0 and 0.05 (the library route has no in-kernel dropout: its 0.05 column is the tensor-op form).
usage: python tools/ab_sublayer.py            (writes a markdown table to stdout)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("tinyllama", "fastmax"), ("tinyllama", "linearmax"), ("llama7b", "fastmax"), ("llama7b", "linearmax")]


def child():
    sys.path.insert(0, ROOT)
    import statistics
    import torch
    from fastmax_experiments_amd.attention_block import CausalSelfAttention, build_rope_cache
    for cfg, alg in CASES:
        n_embd, n_head, groups, B, T = {"tinyllama": (2048, 32, 4, 8, 2048), "llama7b": (4096, 32, 32, 2, 4096)}[cfg]
        for drop in (0.0, 0.05):
            torch.manual_seed(0)
            blk = CausalSelfAttention(n_embd, n_head, n_query_groups=groups, attn_alg=alg, dropout=drop).to(torch.bfloat16)
            torch.nn.init.normal_(blk.attn.lora_B, std=0.02)
            blk.quantize_base().cuda()
            blk.train()
            cos, sin = build_rope_cache(T, blk.rope_n_elem, device="cuda")
            x = torch.randn(B, T, n_embd, device="cuda", dtype=torch.bfloat16, requires_grad=True)
            gy = torch.randn(B, T, n_embd, device="cuda", dtype=torch.bfloat16)

            def step():
                y = blk(x, cos, sin)
                y.backward(gy)
                x.grad = None
                for p in blk.parameters():
                    p.grad = None
            try:
                for _ in range(3):
                    step()
            except NotImplementedError as e:
                print(f"{cfg}|{alg}|{drop}|nan|nan", flush=True)
                continue
            ts = []
            for _ in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    step()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 5)
            print(f"{cfg}|{alg}|{drop}|{statistics.median(ts):.3f}|{min(ts):.3f}", flush=True)
            del blk, x, gy
            torch.cuda.empty_cache()


def main():
    res = {}
    for rep in range(2):                       # two interleaved passes: box drift shows as a difference between them
        for route in ("gemm", "library"):
            env = dict(os.environ, FASTMAX_QLORA_ROUTE=route)
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
            if out.returncode:
                print(out.stdout[-2000:], out.stderr[-2000:])
                sys.exit(out.returncode)
            for line in out.stdout.splitlines():
                if line.count("|") == 4:
                    cfg, alg, drop, med, mn = line.split("|")
                    res.setdefault((cfg, alg, drop), {}).setdefault(route, []).append(float(med))
    print("| sub-layer (fwd+bwd, ms; two passes) | lora_dropout | hand-written route | library route | ratio |")
    print("|---|---|---|---|---|")
    for (cfg, alg, drop), r in res.items():
        g, l = r.get("gemm", []), r.get("library", [])
        fmt = lambda v: " / ".join("–" if x != x else f"{x:.3f}" for x in v)
        ok = g and l and all(x == x for x in g + l)
        ratio = f"{min(l) / min(g):.3f}" if ok else "–"
        print(f"| {cfg} {alg} | {drop} | {fmt(g)} | {fmt(l)} | {ratio} |")


if __name__ == "__main__":
    child() if len(sys.argv) > 1 and sys.argv[1] == "--child" else main()
