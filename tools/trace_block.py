#!/usr/bin/env python3
"""Run the attention sub-layer (QLoRA qkv linear -> RoPE -> fastmax / linearmax -> proj) forward+backward a few times,
for rocprofv3 --kernel-trace --stats.  usage: trace_block.py [tinyllama|llama7b] [fastmax|linearmax] [B] [T]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastmax_experiments_amd.attention_block import CausalSelfAttention, build_rope_cache
cfg = sys.argv[1] if len(sys.argv) > 1 else "tinyllama"
alg = sys.argv[2] if len(sys.argv) > 2 else "linearmax"
n_embd, n_head, groups = {"tinyllama": (2048, 32, 4), "llama7b": (4096, 32, 32)}[cfg]
B = int(sys.argv[3]) if len(sys.argv) > 3 else (8 if cfg == "tinyllama" else 2)
T = int(sys.argv[4]) if len(sys.argv) > 4 else (2048 if cfg == "tinyllama" else 4096)
if os.environ.get("FASTMAX_TUNE_GEMMS"):
    from fastmax_experiments_amd import lora as _lora
    _lora.enable_gemm_tuning()
torch.manual_seed(0)
blk = CausalSelfAttention(n_embd, n_head, n_query_groups=groups, attn_alg=alg).to(torch.bfloat16)
torch.nn.init.normal_(blk.attn.lora_B, std=0.02)
blk.quantize_base().cuda()
blk.group_views = os.environ.get("FASTMAX_GROUP_VIEWS", "1") != "0"      # 0: the reference's expand, materialised (A/B)
cos, sin = build_rope_cache(T, blk.rope_n_elem, device="cuda")
x = torch.randn(B, T, n_embd, device="cuda", dtype=torch.bfloat16, requires_grad=True)
gy = torch.randn(B, T, n_embd, device="cuda", dtype=torch.bfloat16)
import time
for it in range(8):
    if it == 3:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    y = blk(x, cos, sin)
    y.backward(gy)
    x.grad = None
    for p in blk.parameters():
        p.grad = None
torch.cuda.synchronize()
print(f"{cfg} {alg} B={B} T={T}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per fwd+bwd", flush=True)
