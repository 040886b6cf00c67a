"""The operator inside its caller, at the head shapes of the BASELINE.json configs: the HIP path (NF4 + LoRA
linears, fastmax / linearmax) against the same block evaluated with dense float32 tensor math and the CPU
oracle for the attention.  Also one data-parallel fine-tune step (world size 1) on the GPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")


def _reference_block(blk, x, cos, sin):
    """float32 dense evaluation of the same block; the attention through the CPU oracle."""
    from fastmax_experiments_amd import lora
    from fastmax_experiments_amd.attention_block import apply_rope
    from oracle import c_oracle, fastmax_oracle as orc

    def lin(layer, t):
        base = layer.linear
        w = base.dequantize(torch.float32) if isinstance(base, lora.NF4Linear) else base.weight.float()
        y = t @ w.T + (0 if base.bias is None else base.bias.float())
        if layer.r > 0 and hasattr(layer, "lora_A"):
            y = y + (t @ layer.lora_A.float().T) @ layer._dense_rows().float().T * layer.scaling
        return y

    B, T, _ = x.shape
    qkv = lin(blk.attn, x.float())
    q_per_kv = blk.n_head // blk.n_query_groups
    qkv = qkv.view(B, T, blk.n_query_groups, q_per_kv + 2, blk.head_size).permute(0, 2, 3, 1, 4)
    q, k, v = qkv.split((q_per_kv, 1, 1), dim=2)
    k = k.expand(B, blk.n_query_groups, q_per_kv, T, blk.head_size)
    v = v.expand(B, blk.n_query_groups, q_per_kv, T, blk.head_size)
    q, k, v = (t.reshape(B, -1, T, blk.head_size) for t in (q, k, v))
    n = blk.rope_n_elem
    q = torch.cat((apply_rope(q[..., :n], cos.float(), sin.float()), q[..., n:]), -1)
    k = torch.cat((apply_rope(k[..., :n], cos.float(), sin.float()), k[..., n:]), -1)
    qn, kn, vn = (t.cpu().numpy() for t in (q, k, v))
    if blk.attn_alg == "linearmax" and T > 4096:
        # long context: prologue in numpy, then the C oracle's first-order scan (OpenMP over heads, linear in T)
        qq, kk = orc.normalize_qk(qn, kn)
        y, _ = c_oracle.fwd(qq.astype(np.float32), kk.astype(np.float32), vn, mask=True, nt=1.0, p=1)
    elif blk.attn_alg == "linearmax":
        y = orc.linearmax_fwd(qn, kn, vn, chunk=64)
    elif T <= 1024:
        y, _ = c_oracle.fwd(qn, kn, vn, mask=True, p=2)
    else:
        # long sequences: the dense fp64 known-answer form, one head at a time (the factorised second-order state is
        # D^2 (D+1) numbers per token: slower than N^2 D below N ~ 2 D^2)
        y = np.concatenate([orc.fastmax_fwd_dense(qn[:, h:h + 1], kn[:, h:h + 1], vn[:, h:h + 1], mask=True, p=2)[0]
                            for h in range(qn.shape[1])], axis=1)
    y = torch.from_numpy(np.asarray(y, dtype=np.float32)).to(x.device).reshape(B, T, blk.head_size * blk.n_head)
    return lin(blk.proj, y)


@pytest.mark.parametrize("name,T,alg,quant", [
    ("pythia-14m", 1024, "fastmax", False),         # BASELINE config 2 head shape: bf16 forward, seq 1024
    ("pythia-14m", 1024, "linearmax", False),
    ("tiny-llama-1.1b", 2048, "fastmax", True),     # config 3: QLoRA (NF4) + fastmax, seq 2048
    ("Llama-2-7b-hf", 4096, "fastmax", True),       # config 4: QLoRA + fastmax (p=2, D=128), seq 4096
    ("Llama-2-7b-hf", 16384, "linearmax", True),    # config 5: linearmax long context, seq 16384
    ("pythia-1b", 512, "fastmax", False),           # the largest head size of the reference's configs: 8 heads of 256
    ("pythia-1b", 512, "linearmax", True),
    ("Gemma-2b", 512, "fastmax", True),             # multi-query, head size 256: one K / V head viewed by the 8 query heads
])
def test_block_forward_at_config_shapes(name, T, alg, quant):
    from fastmax_experiments_amd.attention_block import CONFIG_SHAPES, CausalSelfAttention, build_rope_cache
    torch.manual_seed(0)
    blk = CausalSelfAttention(attn_alg=alg, r=8, alpha=16, **CONFIG_SHAPES[name])
    torch.nn.init.normal_(blk.attn.lora_B, std=0.02)
    if quant:
        blk.quantize_base()
        blk = blk.cuda()
    else:
        blk = blk.cuda().to(torch.bfloat16)                     # "bf16-true" precision of the reference scripts
    cos, sin = build_rope_cache(T, blk.rope_n_elem, device="cuda")
    x = torch.randn(1, T, CONFIG_SHAPES[name]["n_embd"], device="cuda", dtype=torch.bfloat16)
    with torch.no_grad():
        y = blk(x, cos.to(torch.bfloat16), sin.to(torch.bfloat16))
        ref = _reference_block(blk, x, cos, sin)
    assert y.shape == x.shape and y.dtype == torch.bfloat16
    # bf16 activations end to end (three roundings: qkv, attention, proj) against float32 math on the same weights
    err = rel_err(y.float().cpu().numpy(), ref.cpu().numpy())
    print(f"block forward {name} T={T} {alg}: rel_err {err:.3e}")
    assert err < 1e-2


def test_dp_finetune_step_on_gpu():
    """finetune/lora.py:214-226 step structure on the device: QLoRA block, accumulation, flat-bucket sync, AdamW."""
    from fastmax_experiments_amd import dp, lora
    from fastmax_experiments_amd.attention_block import CONFIG_SHAPES, CausalSelfAttention, build_rope_cache
    torch.manual_seed(0)
    blk = CausalSelfAttention(attn_alg="fastmax", r=8, alpha=16, **CONFIG_SHAPES["tiny-llama-1.1b"]).quantize_base().cuda()
    params = dp.trainable_lora_parameters(blk)
    assert {n for n, p in blk.named_parameters() if p.requires_grad} == {"attn.lora_A", "attn.lora_B"}
    assert blk.attn.linear.weight.dtype == torch.uint8
    T = 256
    cos, sin = (t.to(torch.bfloat16) for t in build_rope_cache(T, blk.rope_n_elem, device="cuda"))
    opt = torch.optim.AdamW(params, lr=1e-3)

    def loss_fn(model, batch):
        x, tgt = batch
        return F.mse_loss(model(x, cos, sin).float(), tgt.float())

    st = dp.DataParallelStepper(blk, opt, dp.TrainArgs(global_batch_size=4, micro_batch_size=2), loss_fn)
    assert st.accum == 2
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(4, T, 2048, device="cuda", generator=g).to(torch.bfloat16)
    tgt = torch.randn(4, T, 2048, device="cuda", generator=g).to(torch.bfloat16)
    before = blk.attn.lora_B.detach().clone()
    l0 = st.micro_step((x[:2], tgt[:2]))
    assert st.step_count == 0 and float(st.bucket.flat.abs().sum()) > 0          # accumulating, no step yet
    st.micro_step((x[2:], tgt[2:]))
    assert st.step_count == 1 and float(st.bucket.flat.abs().sum()) == 0          # stepped and zeroed
    assert not torch.equal(before, blk.attn.lora_B.detach()) and torch.isfinite(l0)
    l1 = loss_fn(blk, (x[:2], tgt[:2]))
    assert torch.isfinite(l1)


@pytest.mark.parametrize("dt,tol", [(torch.float32, 0.0), (torch.bfloat16, 0.0), (torch.float16, 0.0)])
@pytest.mark.parametrize("B,T,G,qpk,hs,rot", [(2, 129, 4, 8, 64, 1.0), (1, 300, 2, 1, 128, 1.0), (2, 64, 3, 2, 64, 0.5), (1, 17, 1, 4, 32, 0.5)])
def test_fused_neighbours_kernel_matches_the_tensor_ops(B, T, G, qpk, hs, rot, dt, tol):
    """fastmax_rope.hip (de-interleave + RoPE + GQA expand, and its backward) against the tensor-op sequence of
    lit_gpt/model.py:397-425: the forward reproduces the reference's roundings and is bit-identical"""
    from fastmax_experiments_amd import ops
    from fastmax_experiments_amd.attention_block import apply_rope, build_rope_cache
    n = int(rot * hs)
    if not ops.rope_qkv_supported(dt, hs, n):
        pytest.skip("piece alignment")
    g = torch.Generator().manual_seed(T + hs)
    qkv = torch.randn(B, T, G, qpk + 2, hs, generator=g).to(dt).cuda()
    gq, gk, gv = (torch.randn(B, G * qpk, T, hs, generator=g).to(dt).cuda() for _ in range(3))
    cos, sin = build_rope_cache(T, n, device="cuda")

    def tensor_ops(x):
        y = x.permute(0, 2, 3, 1, 4)
        q, k, v = y.split((qpk, 1, 1), dim=2)
        k = k.expand(B, G, qpk, T, hs)
        v = v.expand(B, G, qpk, T, hs)
        q, k, v = (t.reshape(B, -1, T, hs) for t in (q, k, v))
        q = torch.cat((apply_rope(q[..., :n], cos, sin), q[..., n:]), -1)
        k = torch.cat((apply_rope(k[..., :n], cos, sin), k[..., n:]), -1)
        return q, k, v

    a = qkv.clone().requires_grad_(True)
    b = qkv.clone().requires_grad_(True)
    qa, ka, va = ops.RopeQKVSplit.apply(a, cos, sin, n)
    qb, kb, vb = tensor_ops(b)
    for x, y in ((qa, qb), (ka, kb), (va, vb)):
        assert x.shape == y.shape and rel_err(x.detach().float().cpu().numpy(), y.detach().float().cpu().numpy()) <= tol
    torch.autograd.backward((qa, ka, va), (gq, gk, gv))
    torch.autograd.backward((qb, kb, vb), (gq, gk, gv))
    # the tensor-op backward rounds the per-head k, v gradients before summing the group; the kernel sums in fp32
    assert rel_err(a.grad.float().cpu().numpy(), b.grad.float().cpu().numpy()) < (1e-6 if dt == torch.float32 else 6e-3)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_fused_neighbours_with_a_16bit_rope_cache(dt):
    """"bf16-true" precision keeps the rope cache in the tensors' dtype: apply_rope (model.py:702-708) then rounds x cos and
    rot(x) sin to 16 bits before adding them.  The kernel reproduces those roundings bit for bit (checked against the
    tensor-op sequence run with 16-bit tables; the reference's own module cannot be imported here: parity of this row is
    pinned only through this restatement)."""
    from fastmax_experiments_amd import ops
    from fastmax_experiments_amd.attention_block import apply_rope, build_rope_cache
    B, T, G, qpk, hs = 2, 131, 2, 4, 64
    g = torch.Generator().manual_seed(9)
    qkv = torch.randn(B, T, G, qpk + 2, hs, generator=g).to(dt).cuda()
    cos, sin = (t.to(dt) for t in build_rope_cache(T, hs, device="cuda"))
    y = qkv.permute(0, 2, 3, 1, 4)
    q, k, v = y.split((qpk, 1, 1), dim=2)
    q = q.reshape(B, -1, T, hs)
    k = k.expand(B, G, qpk, T, hs).reshape(B, -1, T, hs)
    want_q, want_k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
    assert want_q.dtype == dt
    got_q, got_k, _ = ops.RopeQKVSplit.apply(qkv, cos, sin, hs)
    assert torch.equal(got_q, want_q) and torch.equal(got_k, want_k)
    # and float32 tables still give the float32-product result (differs from the 16-bit-table one in the last bit somewhere)
    q32, _, _ = ops.RopeQKVSplit.apply(qkv, cos.float(), sin.float(), hs)
    assert torch.equal(q32, apply_rope(q, cos.float(), sin.float()))


def test_block_with_and_without_fused_neighbours():
    from fastmax_experiments_amd.attention_block import CONFIG_SHAPES, CausalSelfAttention, build_rope_cache
    torch.manual_seed(0)
    cfg = CONFIG_SHAPES["tiny-llama-1.1b"]
    blk = CausalSelfAttention(cfg["n_embd"], cfg["n_head"], n_query_groups=cfg["n_query_groups"], attn_alg="linearmax").to(torch.bfloat16)
    torch.nn.init.normal_(blk.attn.lora_B, std=0.02)
    blk.quantize_base().cuda()
    T = 512
    cos, sin = build_rope_cache(T, blk.rope_n_elem, device="cuda")
    x = torch.randn(2, T, cfg["n_embd"], device="cuda", dtype=torch.bfloat16)
    outs, grads = [], []
    for fused in (True, False):
        blk.fused_neighbours = fused
        xx = x.clone().requires_grad_(True)
        y = blk(xx, cos, sin)
        y.backward(torch.ones_like(y))
        outs.append(y.detach().float().cpu().numpy())
        grads.append(xx.grad.float().cpu().numpy())
    assert rel_err(outs[0], outs[1]) < 5e-3
    assert rel_err(grads[0], grads[1]) < 1e-2


@pytest.mark.parametrize("rep,T", [(4, 600), (3, 256), (5, 256), (6, 256), (7, 256), (3, 4096), (7, 4096), (9, 700)])
def test_grouped_query_prologue_matches_expand_then_normalise(rep, T):
    """fastmax_hack_grouped (K normalised once per key head, the prologue's store writes the per-query-head copies, its backward
    sums a group's gradients) == fastmax_hack on the expanded K, forward and all three gradients; and RopeQKVSplit with
    expand=2 leaves K at its groups with the same values.  Group sizes that do not divide 256 included: the backward's
    per-block records must fit the workspace the host sized (ceil(N / tok) <= rep ceil(N / 256))"""
    from fastmax_experiments_amd import ops
    from fastmax_experiments_amd.attention_mechanisms.fastmax_hack import fastmax_hack, fastmax_hack_grouped
    torch.manual_seed(21)
    B, G, hs = 2, 2, 64
    q0 = torch.randn(B, G * rep, T, hs, device="cuda", dtype=torch.bfloat16)
    kg0 = torch.randn(B, G, T, hs, device="cuda", dtype=torch.bfloat16)
    v0 = torch.randn(B, G * rep, T, hs, device="cuda", dtype=torch.bfloat16)
    go = torch.randn(B, G * rep, T, hs, device="cuda", dtype=torch.bfloat16)
    res = []
    for grouped in (True, False):
        q, kg, v = (t.clone().requires_grad_(True) for t in (q0, kg0, v0))
        if grouped:
            o = fastmax_hack_grouped(q, kg, v, rep, p=1)
        else:
            k = kg[:, :, None].expand(B, G, rep, T, hs).reshape(B, G * rep, T, hs)
            o = fastmax_hack(q, k, v, p=1, mask=True)
        o.backward(go)
        res.append((o.detach(), q.grad, kg.grad, v.grad))
    for a, b in zip(*res):
        assert a.shape == b.shape and float((a.float() - b.float()).abs().max()) <= 2e-2 * float(b.float().abs().max())
    # the values the attention sees are the same bits: normalised copies
    y1, _ = ops.normalize_cast(kg0, rep)
    y2, _ = ops.normalize_cast(kg0[:, :, None].expand(B, G, rep, T, hs).reshape(B, G * rep, T, hs).contiguous())
    assert torch.equal(y1, y2)
    # rope split with K left at its groups
    qkv = torch.randn(B, T, G, rep + 2, hs, device="cuda", dtype=torch.bfloat16)
    from fastmax_experiments_amd.attention_block import build_rope_cache
    cos, sin = build_rope_cache(T, hs, device="cuda")
    qa, ka, va = ops.RopeQKVSplit.apply(qkv, cos, sin, hs, 1)
    qb, kb, vb = ops.RopeQKVSplit.apply(qkv, cos, sin, hs, 2)
    assert kb.shape == (B, G, T, hs) and torch.equal(qa, qb) and torch.equal(va, vb)
    assert torch.equal(ka.view(B, G, rep, T, hs)[:, :, 0], kb) and torch.equal(ka.view(B, G, rep, T, hs)[:, :, rep - 1], kb)


def test_block_grouped_route_matches_the_expanded_route(monkeypatch):
    """CausalSelfAttention (linearmax, grouped-query heads, training): the route that keeps K at its groups through the prologue
    against the same block with that route disabled -- output and gradients"""
    import fastmax_experiments_amd.attention_block as ab
    torch.manual_seed(22)
    blk = ab.CausalSelfAttention(256, 8, n_query_groups=2, attn_alg="linearmax").to(torch.bfloat16)
    torch.nn.init.normal_(blk.attn.lora_B, std=0.02)
    blk.quantize_base().cuda()
    T = 700
    cos, sin = ab.build_rope_cache(T, blk.rope_n_elem, device="cuda")
    x0 = torch.randn(2, T, 256, device="cuda", dtype=torch.bfloat16)
    gy = torch.randn(2, T, 256, device="cuda", dtype=torch.bfloat16)
    res = []
    for grouped in (True, False):
        if not grouped:
            monkeypatch.setattr(ab, "grouped_route_supported", lambda *a: False)
        x = x0.clone().requires_grad_(True)
        y = blk(x, cos, sin)
        y.backward(gy)
        res.append((y.detach(), x.grad, blk.attn.lora_A.grad.clone(), blk.attn.lora_B.grad.clone()))
        for p in blk.parameters():
            p.grad = None
    for a, b in zip(*res):
        assert float((a.float() - b.float()).abs().max()) <= 3e-2 * float(b.float().abs().max())


def _dense_block_fp32(blk, x, cos, sin, A, Bm, p):
    """float32 torch evaluation of the sub-layer with DENSE attention f(QK^T)V (known-answer form, fastmax.py:336-381 + 103)
    on the dequantised weights; A, Bm are float32 leaf copies of attn.lora_A / attn.lora_B so autograd gives their gradients."""
    from fastmax_experiments_amd.attention_block import apply_rope
    B, T, _ = x.shape
    w = blk.attn.linear.dequantize(torch.float32)
    qpk = blk.n_head // blk.n_query_groups
    rows = torch.zeros(blk.attn.linear.out_features, A.shape[0], device=x.device)
    ind = blk.attn._ind.to(x.device)
    rows = rows.index_put((ind[:, None].expand(-1, blk.attn.r), blk.attn._cols.to(x.device)), Bm)
    qkv = x @ w.T + (x @ A.T) @ rows.T * blk.attn.scaling
    qkv5 = qkv.view(B, T, blk.n_query_groups, qpk + 2, blk.head_size)
    q = qkv5[:, :, :, :qpk].permute(0, 2, 3, 1, 4).reshape(B, blk.n_head, T, blk.head_size)
    k, v = (qkv5[:, :, :, qpk + i].permute(0, 2, 1, 3).repeat_interleave(qpk, dim=1) for i in (0, 1))
    n = blk.rope_n_elem
    q = torch.cat((apply_rope(q[..., :n], cos, sin), q[..., n:]), -1)
    k = torch.cat((apply_rope(k[..., :n], cos, sin), k[..., n:]), -1)
    s = (q @ k.transpose(-1, -2)) / (8.0 * blk.head_size ** 0.5)
    P = torch.tril(1 + s + (0.5 * s * s if p == 2 else 0))
    y = (P @ v) / P.sum(-1, keepdim=True)
    y = y.reshape(B, T, blk.head_size * blk.n_head)                     # quirk Q3: no transpose
    return y @ blk.proj.linear.dequantize(torch.float32).T


def test_qlora_block_gradients_against_dense_fp32():
    """gradients of the whole QLoRA attention sub-layer (NF4 linears with LoRA on q, v -> RoPE / GQA split -> fastmax p=2 ->
    proj) w.r.t. the input and both LoRA matrices against float32 autograd through a dense evaluation of the same block"""
    from fastmax_experiments_amd.attention_block import CausalSelfAttention, build_rope_cache
    torch.manual_seed(3)
    blk = CausalSelfAttention(256, 8, n_query_groups=2, attn_alg="fastmax", r=8, alpha=16)
    torch.nn.init.normal_(blk.attn.lora_B, std=0.05)
    blk.quantize_base().cuda()
    T = 320
    cos, sin = build_rope_cache(T, blk.rope_n_elem, device="cuda")
    x0 = torch.randn(2, T, 256, device="cuda")
    gy = torch.randn(2, T, 256, device="cuda")
    x = x0.to(torch.bfloat16).requires_grad_(True)
    y = blk(x, cos.to(torch.bfloat16), sin.to(torch.bfloat16))
    y.backward(gy.to(torch.bfloat16))
    xr = x0.to(torch.bfloat16).float().requires_grad_(True)
    A = blk.attn.lora_A.detach().float().clone().requires_grad_(True)
    Bm = blk.attn.lora_B.detach().float().clone().requires_grad_(True)
    yr = _dense_block_fp32(blk, xr, cos, sin, A, Bm, 2)
    yr.backward(gy.to(torch.bfloat16).float())
    err = rel_err(y.detach().float().cpu().numpy(), yr.detach().cpu().numpy())
    print(f"block fwd rel_err {err:.3e}")
    assert err < 1e-2
    for got, want, name in ((x.grad, xr.grad, "x"), (blk.attn.lora_A.grad, A.grad, "lora_A"), (blk.attn.lora_B.grad, Bm.grad, "lora_B")):
        err = rel_err(got.float().cpu().numpy(), want.cpu().numpy())
        print(f"block grad {name} rel_err {err:.3e}")
        assert err < 1e-2, name


def test_bench_dp_step_on_device():
    """`bench.py --workload dp_step` with the real QLoRA stack on the card: one rank, and two ranks over gloo sharing the one
    GPU (rehearsal of the N > 1 control flow; the RCCL path needs one GPU per rank and runs in the driver's scaling bench)"""
    from test_dp_gloo import _run_bench
    common = ["--workload", "dp_step", "--config", "pythia-14m", "--layers", "2", "--seq", "256", "--steps", "2", "--warmup", "1",
              "--precondition-ms", "0"]
    one = _run_bench(common, 1)
    assert one["n_gpus"] == 1 and one["allreduce"]["backend"].startswith("none") and one["value"] > 0
    assert one["trainable_params"] == 2 * (16 * 128 + 2 * 128 * 8)          # r = 8 on q, v of two layers (A: 16 x 128, B: 256 x 8)
    two = _run_bench(common, 2, env={"FASTMAX_BENCH_BACKEND": "gloo"})
    assert two["n_gpus"] == 2 and two["allreduce"]["backend"] == "gloo" and two["config"]["global_batch"] == 8
    assert two["last_loss"] == two["last_loss"] and two["allreduce"]["bucket_bytes"] == 4 * two["trainable_params"]


@pytest.mark.parametrize("alg", ["fastmax", "linearmax"])
@pytest.mark.parametrize("train", [True, False])
def test_group_views_match_the_expanded_copies_bit_for_bit(alg, train):
    """grouped-query heads with K, V as stride-0 views over (batch x group) -- no per-query-head copy of K or V is ever
    written -- against the same block with the reference's expand (model.py:404-420) materialised: the attention kernels
    see the same values through different strides, so outputs and every gradient are the same bits.  (linearmax with gradients:
    the views take the one-node route that normalises inside the scans, the expanded copies the two-node route that stores
    normalised tensors -- same function, roundings of the row means differ in the last place, so that pair is compared to 1 % of
    the tensor scale instead.)"""
    from fastmax_experiments_amd.attention_block import CONFIG_SHAPES, CausalSelfAttention, build_rope_cache
    torch.manual_seed(31)
    cfg = CONFIG_SHAPES["tiny-llama-1.1b"]
    blk = CausalSelfAttention(cfg["n_embd"], cfg["n_head"], n_query_groups=cfg["n_query_groups"], attn_alg=alg).to(torch.bfloat16)
    torch.nn.init.normal_(blk.attn.lora_B, std=0.02)
    blk.quantize_base().cuda()
    T = 640
    cos, sin = build_rope_cache(T, blk.rope_n_elem, device="cuda")
    x0 = torch.randn(2, T, cfg["n_embd"], device="cuda", dtype=torch.bfloat16)
    gy = torch.randn(2, T, cfg["n_embd"], device="cuda", dtype=torch.bfloat16)
    res = []
    for views in (True, False):
        blk.group_views = views
        for p in blk.parameters():
            p.grad = None
        if train:
            x = x0.clone().requires_grad_(True)
            y = blk(x, cos, sin)
            y.backward(gy)
            res.append((y.detach(), x.grad, blk.attn.lora_A.grad.clone(), blk.attn.lora_B.grad.clone()))
        else:
            with torch.no_grad():
                res.append((blk(x0, cos, sin),))
    for a, b in zip(*res):
        if alg == "linearmax" and train:
            assert float((a.float() - b.float()).abs().max()) <= 1e-2 * float(b.float().abs().max())
        else:
            assert torch.equal(a, b)


def test_rope_split_group_views_are_views():
    from fastmax_experiments_amd import ops
    from fastmax_experiments_amd.attention_block import build_rope_cache
    B, T, G, rep, hs = 2, 70, 3, 4, 64
    qkv = torch.randn(B, T, G, rep + 2, hs, device="cuda", dtype=torch.bfloat16)
    cos, sin = build_rope_cache(T, hs, device="cuda")
    q1, k1, v1 = ops.RopeQKVSplit.apply(qkv, cos, sin, hs, 1)
    q3, k3, v3 = ops.RopeQKVSplit.apply(qkv, cos, sin, hs, 3)
    assert q3.shape == k3.shape == v3.shape == (B * G, rep, T, hs) and k3.stride(1) == 0 and v3.stride(1) == 0
    assert torch.equal(q3.reshape(B, G * rep, T, hs), q1) and torch.equal(k3.reshape(B, G * rep, T, hs), k1)
    assert torch.equal(v3.reshape(B, G * rep, T, hs), v1)
    assert k3.untyped_storage().nbytes() == B * G * T * hs * 2          # one copy per key head, not per query head
    _, k4, v4 = ops.RopeQKVSplit.apply(qkv, cos, sin, hs, 4)
    assert k4.shape == (B, G, T, hs) and v4.stride(1) == 0


def test_flat_grad_bucket_with_bf16_parameters_on_device():
    """bf16 LoRA parameters with the fp32 bucket (the "bf16-true" fine-tune): gradients are copied into the bucket's fp32 views
    at the boundary, accumulate there across micro-batches, and come back in the parameter's dtype; fp32 parameters alias it."""
    from fastmax_experiments_amd import dp
    torch.manual_seed(0)
    a = torch.nn.Parameter(torch.randn(8, 64, device="cuda").to(torch.bfloat16))
    b = torch.nn.Parameter(torch.randn(64, 8, device="cuda"))
    bucket = dp.FlatGradBucket([a, b], dtype=torch.float32)
    assert b.grad is not None and b.grad.data_ptr() == bucket._views[1].data_ptr() and a.grad is None
    x = torch.randn(16, 64, device="cuda")
    want_a = torch.zeros(8, 64, device="cuda")
    want_b = torch.zeros(64, 8, device="cuda")
    for _ in range(3):                                               # three micro-batches, gathered after each
        y = (x.to(torch.bfloat16) @ a.t()).float() @ b.t()
        y.square().mean().backward()
        want_a += a.grad.float()
        bucket.gather()
        assert a.grad is None
    want_b = b.grad.clone()
    bucket.all_reduce_mean()                                          # one rank: no collective, gradients handed back
    assert a.grad.dtype == torch.bfloat16 and a.grad.is_cuda
    assert torch.allclose(a.grad.float(), want_a, rtol=1e-2, atol=1e-3)
    assert torch.equal(b.grad, want_b) and b.grad.data_ptr() == bucket._views[1].data_ptr()
    bucket.zero()
    assert a.grad is None and float(bucket.flat.abs().sum()) == 0.0 and b.grad.data_ptr() == bucket._views[1].data_ptr()


@pytest.mark.parametrize("alg,T", [("fastmax", 256), ("linearmax", 1024)])
def test_dp_micro_batch_replayed_as_hip_graph_matches_eager(alg, T):
    """`DataParallelStepper.capture`: one micro-batch (QLoRA attention stack + lm-head loss, forward + backward into the flat
    bucket) recorded as a HIP graph and replayed with new batches -- same losses and the same parameters after the optimizer
    steps as the eager step (the kernels are deterministic: bitwise equality).  linearmax at 1024 tokens: the one-node route with
    the prologue and its backward inside the linear-time scans (statistics on the state pass, row fix-ups) is what gets recorded."""
    from fastmax_experiments_amd import dp, finetune_step
    from fastmax_experiments_amd.attention_block import build_rope_cache
    dev = torch.device("cuda")
    mb, accum = 2, 2
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(2 * accum, mb, T, 128, device=dev, generator=g).to(torch.bfloat16)
    tgt = torch.randint(0, 512, (2 * accum, mb, T), device=dev, generator=g)
    finals, losses = [], []
    for graph in (False, True):
        torch.manual_seed(0)
        model = finetune_step.AttentionStack("pythia-14m", 2, alg, vocab=512, lora_dropout=0.0).prepare(dev)   # no dropout: bitwise
        cos, sin = (t.to(torch.bfloat16) for t in build_rope_cache(T, model.rope_n_elem, device=dev))
        params = dp.trainable_lora_parameters(model)
        opt = torch.optim.AdamW(params, lr=1e-3)
        st = dp.DataParallelStepper(model, opt, dp.TrainArgs(global_batch_size=mb * accum, micro_batch_size=mb),
                                    lambda m, b: m.loss(b[0], b[1], cos, sin))
        if graph:
            st.capture((x[0], tgt[0]))
            assert float(st.bucket.flat.abs().sum()) == 0.0          # warm-up and capture left nothing in the bucket
        ls = [float(st.micro_step((x[i], tgt[i]))) for i in range(2 * accum)]
        assert st.step_count == 2
        losses.append(ls)
        finals.append(torch.cat([p.detach().float().reshape(-1) for p in params]).clone())
    assert losses[0] == losses[1]
    assert torch.equal(finals[0], finals[1])


def test_lora_dropout_draws_a_new_mask_on_every_replay_of_a_captured_step():
    """the dropout seed of the hand-written route is a device tensor drawn from torch's device generator inside the step: a
    captured micro-batch replayed twice on the SAME batch sees two different masks (and so two different losses), while the
    kernels themselves stay deterministic"""
    from fastmax_experiments_amd import dp, finetune_step
    from fastmax_experiments_amd.attention_block import build_rope_cache
    dev = torch.device("cuda")
    T, mb = 2048, 1                                                   # 2048 rows: the tile-GEMM route with in-kernel dropout
    torch.manual_seed(0)
    model = finetune_step.AttentionStack("pythia-14m", 1, "fastmax", vocab=512, lora_dropout=0.5).prepare(dev)
    model.train()
    cos, sin = (t.to(torch.bfloat16) for t in build_rope_cache(T, model.rope_n_elem, device=dev))
    params = dp.trainable_lora_parameters(model)
    opt = torch.optim.SGD(params, lr=0.0)
    st = dp.DataParallelStepper(model, opt, dp.TrainArgs(global_batch_size=4 * mb, micro_batch_size=mb), lambda m, b: m.loss(b[0], b[1], cos, sin))
    g = torch.Generator(device=dev).manual_seed(6)
    x = torch.randn(mb, T, 128, device=dev, generator=g).to(torch.bfloat16)
    tgt = torch.randint(0, 512, (mb, T), device=dev, generator=g)
    st.capture((x, tgt))
    assert st.accum == 4
    st.micro_step((x, tgt))                                           # accumulating: the bucket holds this replay's gradient
    g1 = st.bucket.flat.clone()
    st.micro_step((x, tgt))
    g2 = st.bucket.flat - g1
    assert float(g1.abs().sum()) > 0 and torch.isfinite(g2).all()
    # same batch, same deterministic kernels: equal gradients would mean the replay reused the recorded mask
    assert not torch.allclose(g2, g1, rtol=1e-3, atol=0.0)


@pytest.mark.parametrize("config,T,alg,tables16", [("tiny-llama-1.1b", 2048, "fastmax", False), ("tiny-llama-1.1b", 2048, "linearmax", False),
                                                   ("Llama-2-7b-hf", 1024, "fastmax", False), ("tiny-llama-1.1b", 1024, "linearmax", True)])
def test_one_kernel_qkv_projection_matches_the_separate_passes(config, T, alg, tables16):
    """the qkv projection with the de-interleave and RoPE in its tile epilogue (fastmax_hip_qlora_gemm_rope) against the same
    layer with the separate QKV-split + RoPE pass: the epilogue rotates the bf16-rounded outputs exactly like the separate
    kernel, so the sub-layer's output and every gradient must be bit-identical"""
    from fastmax_experiments_amd.attention_block import CONFIG_SHAPES, CausalSelfAttention, build_rope_cache
    torch.manual_seed(3)
    blk = CausalSelfAttention(attn_alg=alg, r=8, alpha=16, **CONFIG_SHAPES[config]).to(torch.bfloat16)
    torch.nn.init.normal_(blk.attn.lora_B, std=0.02)
    blk.quantize_base().cuda()
    B = 8 if T == 1024 else 4
    cos, sin = build_rope_cache(T, blk.rope_n_elem, device="cuda")
    if tables16:                                                     # a "bf16-true" rope cache: products rounded before the sum
        cos, sin = cos.to(torch.bfloat16), sin.to(torch.bfloat16)
    x = torch.randn(B, T, blk.attn.linear.in_features, device="cuda", dtype=torch.bfloat16)
    gy = torch.randn_like(x)
    res = []
    for one_kernel in (True, False):
        blk.gemm_rope = one_kernel
        xa = x.clone().requires_grad_(True)
        assert blk._one_kernel_qkv(xa, None, B, T, blk.n_head // blk.n_query_groups) == one_kernel
        y = blk(xa, cos, sin)
        y.backward(gy)
        res.append((y.detach().clone(), xa.grad.clone(), blk.attn.lora_A.grad.clone(), blk.attn.lora_B.grad.clone()))
        blk.attn.lora_A.grad = blk.attn.lora_B.grad = None
    blk.gemm_rope = True
    for a, b in zip(*res):
        assert torch.equal(a, b)
