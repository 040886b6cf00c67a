#!/usr/bin/env python3
"""Golden vectors for the operator's neighbours, by RUNNING THE REFERENCE's own pure-torch code.

    python tests/golden/make_golden_neighbours.py          (build container only: /root/reference exists only here)

`lit_gpt` cannot be imported as a package (lit_gpt/__init__.py, model.py and utils.py pull `lightning`,
`fast_transformers`, `fastmax_cuda` ..., none of them installed), but the pieces this path uses are plain torch:

  * lit_gpt/model.py   build_rope_cache (677-701), apply_rope (702-708)
  * lit_gpt/utils.py   chunked_cross_entropy (228-272)
  * lit_gpt/lora.py    LoRALayer, LoRALinear, LoRAQKVLinear (64-433)

Each definition is located by name in the file's syntax tree and executed, as it stands in the file, in a namespace that
holds nothing but `torch`, `torch.nn`, `torch.nn.functional`, `math` and the `typing` names its annotations mention.  No
package import, no stand-in module, no edited or copied source: only the inputs and the outputs of those calls are stored
(.npz, with the seed and the call arguments in `meta`).  Tests read the .npz files only.
"""
import ast
import json
import math
import os
import typing

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/lit_gpt"


def extract(path, names):
    """definitions `names` (functions / classes at module level) of `path`, executed in a bare torch namespace"""
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    picked = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    missing = set(names) - {n.name for n in picked}
    if missing:
        raise RuntimeError(f"{path}: {sorted(missing)} not found")
    ns = {"torch": torch, "nn": nn, "F": F, "math": math}
    ns.update({k: getattr(typing, k) for k in ("Any", "Dict", "List", "Optional", "Tuple", "Type", "Union", "TypeVar", "Mapping")})
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


def save(name, meta, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach()
            if v.dtype in (torch.bfloat16, torch.float16):
                v = v.float()
            v = v.numpy()
        out[k] = v
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.0f} KiB")


def rand(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g, dtype=torch.float32) * scale).to(dtype)


# ---- RoPE ------------------------------------------------------------------------------------------------------------
def rope_cases():
    build_rope_cache, apply_rope = extract(os.path.join(REF, "model.py"), ["build_rope_cache", "apply_rope"])
    # (name, B, heads, T, hs, rope_n_elem, x dtype, cache dtype): model.py:397-425 rotates the first rope_n_elem features of
    # every head of q and k with cos / sin[:T]; the cache is built in fp32 and may be cast ("bf16-true" runs)
    for name, B, H, T, hs, n_elem, xdt, cdt in [
        ("rope_f32_hs64_full", 2, 4, 37, 64, 64, torch.float32, torch.float32),
        ("rope_bf16_hs64_cache_f32", 2, 4, 37, 64, 64, torch.bfloat16, torch.float32),
        ("rope_bf16_hs64_cache_bf16", 2, 4, 37, 64, 64, torch.bfloat16, torch.bfloat16),
        ("rope_bf16_hs128_cache_bf16", 1, 3, 50, 128, 128, torch.bfloat16, torch.bfloat16),
        ("rope_bf16_hs32_partial8", 2, 4, 33, 32, 8, torch.bfloat16, torch.float32),      # pythia: rotary_percentage 0.25
        ("rope_f16_hs64_partial32", 1, 2, 20, 64, 32, torch.float16, torch.float32),
    ]:
        cos, sin = build_rope_cache(seq_len=T + 11, n_elem=n_elem)
        cos, sin = cos.to(cdt), sin.to(cdt)
        x = rand((B, H, T, hs), 100 + T + hs, dtype=xdt)
        roped = apply_rope(x[..., :n_elem], cos[:T], sin[:T])
        y = torch.cat((roped, x[..., n_elem:]), dim=-1)                      # model.py:424-425
        save(name, dict(fn="apply_rope", B=B, H=H, T=T, hs=hs, rope_n_elem=n_elem, x_dtype=str(xdt), cache_dtype=str(cdt),
                        out_dtype=str(y.dtype), seed=100 + T + hs),
             x=x, cos=cos.float(), sin=sin.float(), y=y)


# ---- chunked cross entropy ---------------------------------------------------------------------------------------------
def ce_cases():
    (chunked_cross_entropy,) = extract(os.path.join(REF, "utils.py"), ["chunked_cross_entropy"])
    B, T, V = 3, 40, 97
    logits = rand((B, T, V), 7, 2.0)
    g = torch.Generator().manual_seed(8)
    targets = torch.randint(0, V, (B, T), generator=g)
    masked = targets.clone()
    masked[0, :9] = -1
    masked[2, 30:] = -1
    all_masked = torch.full_like(targets, -1)
    for name, tg, ii in [("plain", targets, -1), ("ignore", masked, -1), ("all_ignored", all_masked, -1), ("ignore_index_5", targets, 5)]:
        out = {}
        for tag, lg, cs in [("tensor_chunk128", logits, 128), ("tensor_chunk16", logits, 16), ("tensor_chunk0", logits, 0),
                            ("list16_chunk128", list(logits.split(16, dim=1)), 128), ("list8_chunk0", list(logits.split(8, dim=1)), 0)]:
            lgr = [l.clone().requires_grad_(True) for l in lg] if isinstance(lg, list) else lg.clone().requires_grad_(True)
            loss = chunked_cross_entropy(lgr, tg, chunk_size=cs, ignore_index=ii)
            out["loss_" + tag] = loss.detach()
            if torch.isfinite(loss) and loss.requires_grad:
                loss.backward()
                grad = torch.cat([l.grad for l in lgr], dim=1) if isinstance(lgr, list) else lgr.grad
                out["dlogits_" + tag] = grad
        save("ce_" + name, dict(fn="chunked_cross_entropy", B=B, T=T, V=V, ignore_index=ii), logits=logits, targets=tg, **out)


# ---- LoRA layers -------------------------------------------------------------------------------------------------------
def lora_cases():
    LoRALayer, LoRALinear, LoRAQKVLinear = extract(os.path.join(REF, "lora.py"), ["LoRALayer", "LoRALinear", "LoRAQKVLinear"])

    def fill(layer, seed):
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for p in layer.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)

    def run(name, layer, x, meta):
        fill(layer, meta["seed"])
        layer.eval()
        xr = x.clone().requires_grad_(True)
        y = layer(xr)
        gy = rand(tuple(y.shape), meta["seed"] + 1)
        y.backward(gy)
        snap = lambda v: v.detach().clone()                         # merge() below writes the weight in place
        arrays = dict(x=x, y=snap(y), gy=gy, dx=snap(xr.grad), weight=snap(layer.linear.weight), lora_A=snap(layer.lora_A),
                      lora_B=snap(layer.lora_B), d_lora_A=snap(layer.lora_A.grad), d_lora_B=snap(layer.lora_B.grad))
        if layer.linear.bias is not None:
            arrays["bias"] = snap(layer.linear.bias)
        if hasattr(layer, "lora_ind"):
            arrays["lora_ind"] = torch.as_tensor(layer.lora_ind)
        if hasattr(layer, "get_lora_AB"):
            arrays["lora_AB"] = snap(layer.get_lora_AB())
        layer.merge()
        arrays["merged_weight"] = snap(layer.linear.weight)
        arrays["y_merged"] = snap(layer(x))
        save(name, meta, **arrays)

    run("lora_linear_r4", LoRALinear(24, 40, r=4, lora_alpha=8, bias=True), rand((2, 5, 24), 21),
        dict(fn="LoRALinear", in_features=24, out_features=40, r=4, lora_alpha=8, bias=True, seed=20))
    # (name, embd, n_head, n_query_groups, enable_lora): MHA / GQA / MQA x which of q, k, v carry LoRA (lit_gpt/lora.py:180-433;
    # the reference's own tests use the same head layouts, tests/test_lora.py:83-159)
    for name, embd, nh, ng, en in [
        ("mha_qv", 32, 4, 4, (True, False, True)), ("mha_qkv", 32, 4, 4, (True, True, True)), ("mha_k", 32, 4, 4, (False, True, False)),
        ("gqa_qv", 32, 4, 2, (True, False, True)), ("gqa_qkv", 32, 4, 2, (True, True, True)), ("gqa_v", 32, 4, 2, (False, False, True)),
        ("mqa_qv", 32, 4, 1, (True, False, True)), ("mqa_qk", 32, 4, 1, (True, True, False)),
    ]:
        hs = embd // nh
        out_features = (nh + 2 * ng) * hs
        layer = LoRAQKVLinear(embd, out_features, n_head=nh, n_query_groups=ng, r=4, lora_alpha=8, enable_lora=en, bias=(ng != 1))
        run("lora_qkv_" + name, layer, rand((2, 6, embd), 31 + nh + ng),
            dict(fn="LoRAQKVLinear", in_features=embd, out_features=out_features, n_head=nh, n_query_groups=ng, r=4, lora_alpha=8,
                 enable_lora=list(en), bias=(ng != 1), seed=30 + nh + ng))


def main():
    torch.set_num_threads(4)
    torch.manual_seed(0)
    rope_cases()
    ce_cases()
    lora_cases()


if __name__ == "__main__":
    main()
