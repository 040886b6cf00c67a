#!/usr/bin/env python3
"""Randomised hunt over the QLoRA layers: random widths, head groupings, ranks, enable patterns, row counts (gemv, 128-tile,
128x256- and 256x256-tile routes), plain and double-quantised scales -- forward and every gradient against float32 math on
the dequantised base weight (the tensor-op form of lit_gpt/lora.py:419-433, as the layer's own conv1d / zero_pad state it).
usage: fuzz_qlora.py [cases] [seed]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from fastmax_experiments_amd import lora


def dense_reference(layer, x):
    wd = layer.linear.dequantize(torch.float32)
    pre = x.float() @ wd.T + (0 if layer.linear.bias is None else layer.linear.bias.float())
    after_A = F.linear(x.float(), layer.lora_A.float())
    if isinstance(layer, lora.LoRAQKVLinear):
        after_B = layer.conv1d(after_A.transpose(-2, -1), layer.lora_B.float().unsqueeze(-1)).transpose(-2, -1)
        return pre + layer.zero_pad(after_B) * layer.scaling
    return pre + after_A @ layer.lora_B.float().T * layer.scaling


def rel(a, b):
    return float((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-20))


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = random.Random(seed)
    bad = 0
    for ci in range(ncases):
        torch.manual_seed(seed * 1000 + ci)
        hs = rng.choice([32, 64, 128])
        groups = rng.choice([1, 2, 4])
        qpk = rng.choice([1, 2, 4])
        n_head = groups * qpk
        if ((n_head + 2 * groups) * hs) % 64:              # the kernels' contract: out_features % 64 == 0 (raises otherwise)
            hs = 64
        n_embd = rng.choice([128, 256, 384, 512, 1024])
        r = rng.choice([4, 8, 16, 32])
        dq = rng.random() < 0.4
        bias = rng.random() < 0.5
        rows = rng.choice([1, 3, 16, 17, 100, 640, 1000, 2048, 2100, 4096, 5000])
        if rng.random() < 0.35:
            out = rng.choice([128, 192, 320, 512, 1024, 2048])        # widths: in % 128 == 0, out % 64 == 0 (the kernels' contract)
            layer = lora.LoRALinear(n_embd, out, r=r, lora_alpha=2 * r, bias=bias)
            desc = f"LoRALinear({n_embd}->{out}, r={r})"
        else:
            enable = rng.choice([(True, False, True), (True, True, True), (False, False, True), (True, False, False)])
            layer = lora.LoRAQKVLinear(n_embd, (n_head + 2 * groups) * hs, n_head=n_head, n_query_groups=groups, r=r, lora_alpha=2 * r,
                                       enable_lora=enable, bias=bias)
            desc = f"LoRAQKVLinear({n_embd}->{(n_head + 2 * groups) * hs}, heads {n_head}/{groups} x {hs}, r={r}, enable={enable})"
        torch.nn.init.normal_(layer.lora_B, std=0.05)
        desc = f"case {ci}: {desc} bias={bias} dq={dq} rows={rows}"
        try:
            layer.quantize_base(double_quant=dq) if dq else layer.quantize_base()
            layer.cuda()
            lora.mark_only_lora_as_trainable(layer)
            x = torch.randn(1, rows, n_embd, device="cuda").to(torch.bfloat16).requires_grad_(True)   # (batch, tokens, width)
            y = layer(x)
            gy = torch.randn_like(y)
            y.backward(gy)
            xr = x.detach().clone().requires_grad_(True)
            A0, B0 = layer.lora_A.grad.clone(), layer.lora_B.grad.clone()
            layer.lora_A.grad = layer.lora_B.grad = None
            yr = dense_reference(layer, xr)
            yr.backward(gy.float())
            errs = {"y": rel(y, yr), "dx": rel(x.grad, xr.grad), "dA": rel(A0, layer.lora_A.grad), "dB": rel(B0, layer.lora_B.grad)}
        except Exception as e:                            # noqa: BLE001
            print("RAISED", desc, type(e).__name__, str(e)[:300], flush=True)
            bad += 1
            continue
        ok = max(errs.values()) < 3e-2
        bad += 0 if ok else 1
        print("ok  " if ok else "BAD ", desc, " ".join(f"{k}={v:.2e}" for k, v in errs.items()), flush=True)
    print(f"{ncases - bad} / {ncases} within tolerance", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
