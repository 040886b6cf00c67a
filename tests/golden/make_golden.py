#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run once in the build container (the only place /root/reference exists):

    python tests/golden/make_golden.py

It imports /root/reference/attention_mechanisms/{fastmax,fastmax_hack}.py under alias
module names (the reference text itself is never copied), feeds seeded inputs, and stores
inputs + the reference's outputs as .npz.  The GPU box and the test-suite only ever read
the .npz files.  Every file records its seed and call arguments in the ``meta`` entry.

Conventions
* main parity sets run the reference in float64 on float32-representable inputs (so the
  very same inputs can be handed to fp32 / bf16 kernels); outputs stored as float64
* dtype-rule sets run it in fp32 / bf16 / fp16 and record the OUTPUT DTYPE it produced
"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/attention_mechanisms"


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


ref = _load("_ref_fastmax", os.path.join(REF, "fastmax.py"))
ref_hack = _load("_ref_fastmax_hack", os.path.join(REF, "fastmax_hack.py"))


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g, dtype=torch.float32) * scale)


def _save(name, meta, **arrays):
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
    np.savez(path, **out)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.0f} KiB")


def fwd_bwd_case(name, shape, p, mask, seed, scale=1.0, **kw):
    B, H, N, D = shape
    q32, k32, v32 = (_rand(shape, seed + i, scale if i < 2 else 1.0) for i in range(3))
    go32 = _rand(shape, seed + 3)
    q, k, v = (t.double().requires_grad_(True) for t in (q32, k32, v32))
    o = ref.fastmax(q, k, v, mask=mask, p=p, **kw)
    # the denominator the reference saves for backward (fastmax.py:106)
    nt = 1 if kw.get("tensors_normalized") else kw.get("normalize_term", 8) * np.sqrt(D)
    with torch.no_grad():
        g = (ref.fastattention_einops.compute_g_masked(q, k, nt, p) if mask
             else ref.fastattention_einops.compute_g_unmasked(q, k, nt, p))
    o.backward(go32.double())
    meta = dict(fn="fastmax", shape=list(shape), p=p, mask=mask, seed=seed, scale=scale,
                kw=kw, ref_dtype="float64")
    _save(name, meta, q=q32, k=k32, v=v32, grad_o=go32, o=o, g=g.double(),
          dq=q.grad, dk=k.grad, dv=v.grad)


def main():
    torch.set_num_threads(8)
    # ---- fixture #1: BASELINE config 1 exactly (torch.manual_seed(0), randn, fp32)
    torch.manual_seed(0)
    q, k, v = (torch.randn(2, 4, 256, 64) for _ in range(3))
    o = ref.fastmax(q, k, v)
    _save("c1_fastmax_p1_masked_fp32", dict(fn="fastmax", shape=[2, 4, 256, 64], p=1,
          mask=True, seed="torch.manual_seed(0); randn x3 (q,k,v)", ref_dtype="float32"),
          q=q, k=k, v=v, o=o)

    # ---- main fwd+bwd parity sets (fp64 reference runs)
    shapes = [(2, 3, 1, 16), (2, 3, 7, 16), (1, 2, 64, 16),
              (2, 3, 7, 32), (1, 2, 64, 32), (1, 1, 256, 32),
              (2, 3, 1, 64), (2, 3, 7, 64), (1, 2, 64, 64), (1, 1, 256, 64),
              (1, 1, 257, 64), (1, 2, 33, 128), (1, 1, 130, 128)]
    seed = 1000
    for shape in shapes:
        for p in (1, 2):
            for mask in (True, False):
                B, H, N, D = shape
                if p == 2 and N * D ** 3 * B * H * 8 > 6e9:
                    continue
                seed += 10
                nm = f"fm_B{B}H{H}N{N}D{D}_p{p}_{'masked' if mask else 'unmasked'}"
                fwd_bwd_case(nm, shape, p, mask, seed)

    # ---- option cases
    fwd_bwd_case("opt_tensors_normalized_p1", (1, 2, 40, 32), 1, True, 5000, scale=0.2,
                 tensors_normalized=True)
    fwd_bwd_case("opt_tensors_normalized_p2", (1, 2, 40, 32), 2, True, 5010, scale=0.2,
                 tensors_normalized=True)
    fwd_bwd_case("opt_normalize_term_3_p1", (1, 2, 40, 64), 1, True, 5020, normalize_term=3.0)
    fwd_bwd_case("opt_normalize_term_3_p2_unmasked", (1, 2, 40, 64), 2, False, 5030,
                 normalize_term=3.0)
    # stress: large scores (g can approach 0 for p=1) -- recorded, loose tolerance in tests
    fwd_bwd_case("stress_qk_scale4_p1", (1, 2, 64, 64), 1, True, 5040, scale=4.0)
    fwd_bwd_case("stress_qk_scale4_p2", (1, 2, 64, 64), 2, True, 5050, scale=4.0)

    # ---- non-contiguous (GQA-expanded K,V as lit_gpt/model.py:411-420 builds them)
    B, G, QPK, N, D = 2, 2, 3, 24, 32
    q = _rand((B, G * QPK, N, D), 6000)
    kg, vg = _rand((B, G, 1, N, D), 6001), _rand((B, G, 1, N, D), 6002)
    k = kg.expand(B, G, QPK, N, D).reshape(B, G * QPK, N, D)
    v = vg.expand(B, G, QPK, N, D).reshape(B, G * QPK, N, D)
    qt = q.transpose(1, 2).contiguous().transpose(1, 2)          # (B,H,N,D) view, N-major
    o = ref.fastmax(qt.double(), k.double(), v.double(), mask=True, p=1)
    _save("noncontig_gqa_p1", dict(fn="fastmax", p=1, mask=True, note="q is a transposed "
          "view; k,v are GQA-expanded; stored here materialised"), q=q, k=k, v=v, o=o)

    # ---- unmasked, N_q != N_k (KV-cache decode, model.py:427-430,464-466); forward only
    for nq, nk, p, sd in ((1, 16, 1, 6100), (1, 16, 2, 6110), (5, 16, 2, 6120)):
        q, k, v = _rand((2, 3, nq, 32), sd), _rand((2, 3, nk, 32), sd + 1), _rand((2, 3, nk, 32), sd + 2)
        o = ref.fastmax(q.double(), k.double(), v.double(), mask=False, p=p)
        _save(f"decode_Nq{nq}_Nk{nk}_p{p}", dict(fn="fastmax", p=p, mask=False), q=q, k=k, v=v, o=o)

    # ---- linearmax = fastmax_hack
    for nm, shape, mask, p, sd in (("hack_masked_p1", (2, 3, 50, 32), True, 1, 7000),
                                   ("hack_masked_p2", (1, 2, 20, 16), True, 2, 7010),
                                   ("hack_masked_p1_D128", (1, 2, 70, 128), True, 1, 7020),
                                   ("hack_unmasked", (2, 3, 50, 32), False, 1, 7030)):
        q, k, v = (_rand(shape, sd + i) for i in range(3))
        o = ref_hack.fastmax_hack(q.double(), k.double(), v.double(), p=p, mask=mask)
        _save(nm, dict(fn="fastmax_hack", p=p, mask=mask, shape=list(shape)), q=q, k=k, v=v, o=o)
    q, k, v = _rand((2, 3, 4, 32), 7100), _rand((2, 3, 16, 32), 7101), _rand((2, 3, 16, 32), 7102)
    o = ref_hack.fastmax_hack(q.double(), k.double(), v.double(), p=1, mask=False)
    _save("hack_unmasked_Nq4_Nk16", dict(fn="fastmax_hack", p=1, mask=False), q=q, k=k, v=v, o=o)
    # linearmax gradients come from plain autograd through the reference graph
    q32, k32, v32, go = (_rand((1, 2, 40, 32), 7200 + i) for i in range(4))
    q, k, v = (t.double().requires_grad_(True) for t in (q32, k32, v32))
    o = ref_hack.fastmax_hack(q, k, v, p=1, mask=True)
    o.backward(go.double())
    _save("hack_masked_p1_grads", dict(fn="fastmax_hack", p=1, mask=True), q=q32, k=k32, v=v32,
          grad_o=go, o=o, dq=q.grad, dk=k.grad, dv=v.grad)

    # ---- normalize()
    q, k = _rand((2, 3, 30, 16), 7300), _rand((2, 3, 30, 16), 7301)
    qn, kn = ref.fastattention_einops.normalize(q.double(), k.double())
    _save("normalize", dict(fn="normalize"), q=q, k=k, qn=qn, kn=kn)

    # ---- create_attn=True (dense known-answer path), forward only
    for p, mask, sd in ((1, True, 7400), (2, True, 7410), (2, False, 7420)):
        q, k, v = (_rand((1, 2, 12, 16), sd + i) for i in range(3))
        o, a = ref.fastmax(q.double(), k.double(), v.double(), mask=mask, p=p, create_attn=True)
        _save(f"create_attn_p{p}_{'masked' if mask else 'unmasked'}",
              dict(fn="fastmax", p=p, mask=mask, create_attn=True), q=q, k=k, v=v, o=o, a=a)

    # ---- dtype rules: record output dtype + values for low-precision runs
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        for mask in (True, False):
            q, k, v = (_rand((1, 2, 32, 32), 8000 + i).to(dt) for i in range(3))
            o = ref.fastmax(q, k, v, mask=mask, p=1)
            nm = f"dtype_{str(dt).split('.')[-1]}_{'masked' if mask else 'unmasked'}"
            _save(nm, dict(fn="fastmax", p=1, mask=mask, in_dtype=str(dt), out_dtype=str(o.dtype)),
                  q=q, k=k, v=v, o=o)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference not present: goldens can only be (re)generated in the build container")
    main()
