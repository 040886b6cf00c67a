"""NF4 + LoRA fused linear on the MI355X: the HIP kernels (through ctypes -> C ABI) against dense tensor
math on the SAME dequantised weights.  bitsandbytes is absent, so NF4 values are 'parity unpinned'
(SURVEY 8c): these tests pin self-consistency -- quantise -> fused kernel == dequantise -> dense matmul."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")


def _rel(a, b):
    return float((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-30))


def test_hip_dequantize_matches_codec_bit_for_bit():
    from fastmax_experiments_amd import lora
    g = torch.Generator().manual_seed(0)
    w = torch.randn(192, 256, generator=g)
    packed, absmax = lora.nf4_quantize(w)
    host = lora.nf4_dequantize(packed, absmax, w.shape)
    dev = lora.nf4_dequantize(packed.cuda(), absmax.cuda(), w.shape, torch.float32)
    assert torch.equal(dev.cpu(), host)
    dev16 = lora.nf4_dequantize(packed.cuda(), absmax.cuda(), w.shape, torch.bfloat16)
    assert torch.equal(dev16.cpu(), host.to(torch.bfloat16))


@pytest.mark.parametrize("M,K,N", [(300, 256, 384), (128, 128, 64), (1000, 512, 1088), (77, 2048, 2560), (1, 4096, 4096), (5, 1408, 320),
                                   (16, 2048, 2560)])          # M <= 16: the column-sliced generation kernel
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_nf4_linear_forward_and_dx(M, K, N, dt):
    from fastmax_experiments_amd import lora
    g = torch.Generator().manual_seed(M + N)
    lin = torch.nn.Linear(K, N)
    torch.nn.init.normal_(lin.weight, generator=g)
    q = lora.NF4Linear.from_linear(lin).cuda()
    x = torch.randn(M, K, generator=g).to(dt).cuda().requires_grad_(True)
    y = q(x)
    assert y.shape == (M, N) and y.dtype == dt
    wd = q.dequantize(torch.float32)
    xr = x.detach().float().to(torch.bfloat16).float()          # the kernel feeds bf16 operands to the matrix cores
    ref = xr @ wd.to(torch.bfloat16).float().T + q.bias.float()
    assert _rel(y, ref) < (1.5e-2 if dt == torch.bfloat16 else 2e-5 * K ** 0.5)
    gy = torch.randn(M, N, generator=g).to(dt).cuda()
    y.backward(gy)
    dref = gy.float().to(torch.bfloat16).float() @ wd.to(torch.bfloat16).float()
    assert x.grad.dtype == dt and _rel(x.grad, dref) < (1.5e-2 if dt == torch.bfloat16 else 2e-5 * N ** 0.5)


def _dense_reference(layer, x):
    """lit_gpt/lora.py:419-433 with the dequantised base weight (float32 math)."""
    from fastmax_experiments_amd import lora
    wd = layer.linear.dequantize(torch.float32)
    pre = x.float() @ wd.T + (0 if layer.linear.bias is None else layer.linear.bias.float())
    after_A = F.linear(x.float(), layer.lora_A.float())
    if isinstance(layer, lora.LoRAQKVLinear):
        after_B = layer.conv1d(after_A.transpose(-2, -1), layer.lora_B.float().unsqueeze(-1)).transpose(-2, -1)
        return pre + layer.zero_pad(after_B) * layer.scaling
    return pre + after_A @ layer.lora_B.float().T * layer.scaling


@pytest.mark.parametrize("kind", ["linear", "qkv_mha", "qkv_gqa_qv", "qkv_gqa_all"])
def test_fused_qlora_layer_forward_backward(kind):
    from fastmax_experiments_amd import lora
    torch.manual_seed(1)
    if kind == "linear":
        layer = lora.LoRALinear(256, 384, r=8, lora_alpha=16)
    elif kind == "qkv_mha":
        layer = lora.LoRAQKVLinear(256, 768, n_head=4, n_query_groups=4, r=8, lora_alpha=16, enable_lora=(True, False, True))
    elif kind == "qkv_gqa_qv":      # TinyLlama-like grouping: 8 heads, 2 KV groups, head 32
        layer = lora.LoRAQKVLinear(256, (8 + 4) * 32, n_head=8, n_query_groups=2, r=8, lora_alpha=16,
                                   enable_lora=(True, False, True))
    else:
        layer = lora.LoRAQKVLinear(256, (8 + 4) * 32, n_head=8, n_query_groups=2, r=8, lora_alpha=16, enable_lora=True)
    torch.nn.init.normal_(layer.lora_B, std=0.05)
    layer.quantize_base().cuda()
    lora.mark_only_lora_as_trainable(layer)
    assert layer.linear.weight.dtype == torch.uint8
    x = torch.randn(2, 75, 256, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    y = layer(x)
    assert y.shape == (2, 75, layer.linear.out_features) and y.dtype == torch.bfloat16
    ref = _dense_reference(layer, x.detach())
    assert _rel(y, ref) < 2e-2
    gy = torch.randn_like(y)
    y.backward(gy)
    # reference gradients by autograd through the dense float32 formulation
    xa = x.detach().float().requires_grad_(True)
    A = layer.lora_A.detach().float().requires_grad_(True)
    B = layer.lora_B.detach().float().requires_grad_(True)
    shadow = type("S", (), {})()
    wd = layer.linear.dequantize(torch.float32)
    after_A = F.linear(xa, A)
    if isinstance(layer, lora.LoRAQKVLinear):
        after_B = layer.conv1d(after_A.transpose(-2, -1), B.unsqueeze(-1)).transpose(-2, -1)
        lo = layer.zero_pad(after_B) * layer.scaling
    else:
        lo = after_A @ B.T * layer.scaling
    yr = xa @ wd.T + layer.linear.bias.float() + lo
    yr.backward(gy.float())
    assert _rel(x.grad, xa.grad) < 3e-2
    assert _rel(layer.lora_A.grad, A.grad) < 3e-2
    assert _rel(layer.lora_B.grad, B.grad) < 3e-2
    assert layer.linear.weight.grad is None


def test_merge_into_nf4_requantises():
    from fastmax_experiments_amd import lora
    torch.manual_seed(2)
    layer = lora.LoRALinear(128, 128, r=4, lora_alpha=8)
    torch.nn.init.normal_(layer.lora_B, std=0.2)
    layer.quantize_base().cuda()
    w0 = layer.linear.dequantize()
    delta = layer.get_lora_AB().float()
    layer.merge()
    assert layer.merged and layer.linear.weight.dtype == torch.uint8
    w1 = layer.linear.dequantize()
    target = w0 + delta
    # merged weight = NF4(w0 + delta): within one quantisation step of the target, and closer to it than w0 was
    assert float((w1 - target).abs().max()) <= 0.16 * float(target.abs().max())
    assert float((w1 - target).norm()) < float((w0 - target).norm())
    x = torch.randn(5, 128, device="cuda", dtype=torch.bfloat16)
    assert _rel(layer(x), x.float() @ w1.T + layer.linear.bias.float()) < 2e-2


def test_rejects_unsupported_shapes_loudly():
    from fastmax_experiments_amd import lora
    q = lora.NF4Linear.from_linear(torch.nn.Linear(64, 64)).cuda()
    with pytest.raises(NotImplementedError):
        q(torch.randn(4, 64, device="cuda", dtype=torch.bfloat16))


def test_dense_weight_cache_gives_the_decode_once_results():
    """NF4Linear.cache_dense(): same values as the decode-once route (the cached matrix IS the decoded weight), forward and
    input / LoRA gradients, at row counts on both sides of DENSE_M; dropped by merge()"""
    import torch
    from fastmax_experiments_amd import lora
    torch.manual_seed(0)
    K, N = 256, 320
    layer = lora.LoRALinear(K, N, r=8, lora_alpha=16, bias=True)
    torch.nn.init.normal_(layer.lora_B, std=0.05)
    layer.quantize_base().cuda()
    for M in (16, 4096):
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        outs = []
        for cached in (False, True):
            layer.linear.cache_dense(cached)
            xx = x.clone().requires_grad_(True)
            y = layer(xx)
            y.float().sum().backward()
            outs.append((y.detach().float(), xx.grad.float(), layer.lora_A.grad.float().clone(), layer.lora_B.grad.float().clone()))
            layer.lora_A.grad = layer.lora_B.grad = None
        for a, b in zip(*outs):
            assert (a - b).abs().max() <= 2e-2 * b.abs().max().clamp_min(1e-6)
    assert lora.cache_dense_weights(layer) == K * N * 2
    layer.merge()
    assert layer.linear._dense_cache is None


@pytest.mark.parametrize("M", [1, 3, 16])
def test_generation_size_rows_with_lora_branch_and_bias(M):
    """M <= 16 takes nf4_gemv_kernel (LoRA k-step and bias in its epilogue): same function as the dense reference"""
    from fastmax_experiments_amd import lora
    torch.manual_seed(M)
    K, N = 1024, 768
    layer = lora.LoRALinear(K, N, r=8, lora_alpha=16, bias=True)
    torch.nn.init.normal_(layer.lora_B, std=0.05)
    torch.nn.init.normal_(layer.linear.bias, std=0.5)
    layer.quantize_base().cuda()
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    with torch.no_grad():
        y = layer(x)
    ref = _dense_reference(layer, x)
    assert y.shape == (M, N) and _rel(y, ref) < 1.5e-2


# ---- rank-r kernels around the library GEMM (csrc/lora_thin.hip) -------------------------------------------------
@pytest.mark.parametrize("M,K,N,R", [(2048, 256, 320, 16), (2049, 512, 128, 8), (4100, 256, 192, 24), (1, 128, 64, 32), (77, 1024, 2560, 16)])
def test_lora_thin_kernels_against_tensor_ops(M, K, N, R):
    """lora_down / lora_tn / lora_up_ against float32 tensor math on the same bf16 operands (ragged row counts, every rank
    the layers can have, bias): down and up round once to bf16, tn sums in float32"""
    from fastmax_experiments_amd import lora
    torch.manual_seed(M + R)
    RP = 16 if R <= 16 else 32
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
    A = torch.zeros(RP, K, device="cuda", dtype=torch.bfloat16)
    A[:R] = torch.randn(R, K, device="cuda") * 0.1
    e, et = lora.lora_down(x, A)
    ref = x.float() @ A.float().t()
    assert _rel(e, ref) < 8e-3
    assert et.shape == (RP, (M + 15) // 16 * 16) and torch.equal(et[:, :M], e.t()) and not et[:, M:].any()
    c = lora.lora_tn(et, dy)
    refc = e.float().t() @ dy.float()
    assert c.dtype == torch.float32 and _rel(c, refc) < 2e-5
    ct = lora.lora_tn(et, dy, R, torch.bfloat16, transpose=True)
    assert ct.shape == (N, R) and ct.dtype == torch.bfloat16 and _rel(ct, refc[:R].t()) < 8e-3
    bn = torch.randn(N, R, device="cuda", dtype=torch.bfloat16) * 0.1
    bias = torch.randn(N, device="cuda")
    y0 = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
    er = e[:, :R].contiguous()
    y = lora.lora_up_(y0.clone(), er, bn, bias)
    assert _rel(y, y0.float() + er.float() @ bn.float().t() + bias) < 8e-3
    y = lora.lora_up_(y0.clone(), er, bn)
    assert _rel(y, y0.float() + er.float() @ bn.float().t()) < 8e-3


@pytest.mark.parametrize("kind", ["linear", "qkv_gqa_qv", "qkv_gqa_all"])
def test_many_rows_route_matches_the_tensor_op_route(kind, monkeypatch):
    """M >= DENSE_M: the HIP rank-r route (_QLoRAThinFn) against the same layer with FASTMAX_LORA_THIN off (library GEMMs
    for every product) and against the dense float32 formulation -- forward, dx, dA, dB"""
    from fastmax_experiments_amd import lora
    torch.manual_seed(3)
    if kind == "linear":
        layer = lora.LoRALinear(256, 384, r=8, lora_alpha=16, bias=True)
    elif kind == "qkv_gqa_qv":
        layer = lora.LoRAQKVLinear(256, (8 + 4) * 32, n_head=8, n_query_groups=2, r=8, lora_alpha=16, enable_lora=(True, False, True))
    else:
        layer = lora.LoRAQKVLinear(256, (8 + 4) * 32, n_head=8, n_query_groups=2, r=8, lora_alpha=16, enable_lora=True)
    torch.nn.init.normal_(layer.lora_B, std=0.05)
    layer.quantize_base().cuda().to(torch.bfloat16)
    lora.mark_only_lora_as_trainable(layer)
    x = torch.randn(3, 700, 256, device="cuda", dtype=torch.bfloat16)
    gy = None
    res = []
    for thin in (True, False):
        monkeypatch.setattr(lora, "LORA_THIN", thin)
        assert lora.thin_route(x, layer.linear) == thin
        xx = x.clone().requires_grad_(True)
        y = layer(xx)
        if gy is None:
            gy = torch.randn_like(y)
        y.backward(gy)
        res.append((y.detach(), xx.grad, layer.lora_A.grad.clone(), layer.lora_B.grad.clone()))
        layer.lora_A.grad = layer.lora_B.grad = None
    for a, b in zip(*res):
        assert a.dtype == b.dtype and _rel(a, b) < 2e-2
    assert _rel(res[0][0], _dense_reference(layer, x)) < 2e-2


def test_many_rows_route_keeps_the_hand_written_kernels_under_dropout(monkeypatch):
    """LoRA dropout is part of the rank-r kernels (round 3): a layer in training mode with lora_dropout > 0 -- the reference's
    default is 0.05, finetune/lora.py:42 -- stays on the hand-written route and hands it its dropout probability"""
    from fastmax_experiments_amd import lora
    torch.manual_seed(4)
    layer = lora.LoRALinear(128, 128, r=8, lora_alpha=16, lora_dropout=0.5)
    layer.quantize_base().cuda().to(torch.bfloat16)
    seen = []
    real = lora.qlora_linear_thin
    monkeypatch.setattr(lora, "qlora_linear_thin", lambda *a, **k: (seen.append(a[5] if len(a) > 5 else k.get("drop_p", 0.0)), real(*a, **k))[1])
    x = torch.randn(2048, 128, device="cuda", dtype=torch.bfloat16)
    layer.train()
    y = layer(x)
    assert seen == [0.5] and y.shape == (2048, 128)
    layer.eval()
    layer(x)
    assert seen == [0.5, 0.0]


def _keep_mask_reference(seed_words, M, K, p):
    """the documented mask function (lora_thin.hip): 16 bits of lowbias32(seed ^ (m * K/2 + k/2)) per element, kept iff >= t"""
    import numpy as np
    t = min(65535, int(p * 65536.0 + 0.5))
    ldw = K // 2
    idx = (np.arange(M, dtype=np.uint64)[:, None] * ldw + np.arange(ldw, dtype=np.uint64)[None, :]) & 0xFFFFFFFF
    x = (idx ^ np.uint64(seed_words[0] & 0xFFFFFFFF)).astype(np.uint64)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x21f0aaad)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x735a2d97)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15)
    keep = np.empty((M, K), dtype=bool)
    keep[:, 0::2] = (x & np.uint64(0xFFFF)) >= t
    keep[:, 1::2] = (x >> np.uint64(16)) >= t
    return keep


@pytest.mark.gpu
@pytest.mark.parametrize("p", [0.05, 0.5])
def test_lora_dropout_mask_is_consistent_forward_and_backward(p):
    """The keep mask is regenerated by three kernels (down: x A^T; tn: dA; up: dx): with the SAME mask taken from
    fastmax_hip_lora_dropout_mask, dense float32 math must reproduce y, dx, dA, dB of the hand-written route -- i.e. forward and
    backward saw one mask -- and the mask has the documented bits and the keep rate 1 - p."""
    import numpy as np
    from fastmax_experiments_amd import lora
    torch.manual_seed(21)
    M, K, N, r = 2304, 256, 384, 8
    layer = lora.LoRALinear(K, N, r=r, lora_alpha=16, lora_dropout=p, bias=True)
    torch.nn.init.normal_(layer.lora_B, std=0.05)
    layer.quantize_base().cuda().to(torch.bfloat16)
    lora.mark_only_lora_as_trainable(layer)
    layer.train()
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    seeds = []
    real_seed = lora.new_dropout_seed
    lora.new_dropout_seed = lambda dev: (seeds.append(real_seed(dev)), seeds[-1])[1]
    try:
        y = layer(x)
    finally:
        lora.new_dropout_seed = real_seed
    assert len(seeds) == 1
    gy = torch.randn_like(y)
    y.backward(gy)
    mask = lora.dropout_mask(seeds[0], M, K, p)
    assert np.array_equal(mask.cpu().numpy(), _keep_mask_reference(seeds[0].cpu().numpy().astype(np.int64), M, K, p))
    rate = float(mask.float().mean())
    assert abs(rate - (1 - p)) < 4 * (p * (1 - p) / (M * K)) ** 0.5 + 2e-5
    # dense float32 reference with that mask
    W = layer.linear.dequantize().float()
    xf = x.detach().float().requires_grad_(True)
    A = layer.lora_A.detach().float().requires_grad_(True)
    B = layer.lora_B.detach().float().requires_grad_(True)
    xd = xf * mask.float() * lora.dropout_scale(p)
    yr = xf @ W.float().t() + layer.linear.bias.float() + (xd @ A.t()) @ B.t() * layer.scaling
    yr.backward(gy.float())
    tol = lambda a: 3e-2 * float(a.abs().max())
    assert float((y.float() - yr).abs().max()) <= tol(yr)
    assert float((x.grad.float() - xf.grad).abs().max()) <= tol(xf.grad)
    assert float((layer.lora_A.grad.float() - A.grad).abs().max()) <= tol(A.grad)
    assert float((layer.lora_B.grad.float() - B.grad).abs().max()) <= tol(B.grad)
    # the branch's share of dx alone (the frozen product removed) must vanish exactly where the mask dropped x
    lora_dx = x.grad.float() - (gy.float() @ W.float())
    dropped = ~mask
    assert float(lora_dx[dropped].abs().max()) <= 2e-2 * float((gy.float() @ W.float()).abs().max())
    # a second call draws another mask
    y2 = layer(x.detach())
    assert not torch.equal(y2, y.detach())


@pytest.mark.parametrize("kind", ["linear", "qkv_gqa_qv", "qkv_gqa_all", "qkv_k_only"])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_scattered_operand_matches_dense_rows(kind, dt):
    """_dense_rows_t (one HIP launch) == (scaling * _dense_rows())^T zero padded to the kernel rank, and its backward is the
    gather of the same entries -- against autograd through the tensor-op formulation"""
    from fastmax_experiments_amd import lora
    torch.manual_seed(7)
    if kind == "linear":
        layer = lora.LoRALinear(128, 192, r=8, lora_alpha=16)
    else:
        en = {"qkv_gqa_qv": (True, False, True), "qkv_gqa_all": True, "qkv_k_only": (False, True, False)}[kind]
        layer = lora.LoRAQKVLinear(128, (8 + 4) * 16, n_head=8, n_query_groups=2, r=8, lora_alpha=16, enable_lora=en)
    torch.nn.init.normal_(layer.lora_B)
    layer.cuda().to(dt)
    et = layer._dense_rows_t()
    R = layer.lora_A.shape[0]
    RP = 16 if R <= 16 else 32
    ref = (layer._dense_rows().float() * layer.scaling).t()
    assert et.shape == (RP, layer.linear.out_features) and et.dtype == torch.bfloat16
    assert torch.equal(et[:R], ref.to(dt).to(torch.bfloat16) if dt == torch.bfloat16 else ref.to(torch.bfloat16)) or _rel(et[:R], ref) < 4e-3
    assert not et[R:].any()
    w = torch.randn_like(et)
    (et.float() * w.float()).sum().backward()
    got = layer.lora_B.grad.clone()
    layer.lora_B.grad = None
    ((layer._dense_rows().float() * layer.scaling).t() * w[:R].float()).sum().backward()
    assert got.dtype == dt and _rel(got, layer.lora_B.grad) < (1e-6 if dt == torch.float32 else 8e-3)


def test_lora_up_transposed_operand():
    from fastmax_experiments_amd import lora
    torch.manual_seed(8)
    M, N, R = 300, 320, 16
    y0 = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
    e = torch.randn(M, R, device="cuda", dtype=torch.bfloat16)
    bn = torch.randn(N, R, device="cuda", dtype=torch.bfloat16) * 0.1
    a = lora.lora_up_(y0.clone(), e, bn)
    b = lora.lora_up_(y0.clone(), e, bn.t().contiguous(), transposed=True)
    assert torch.equal(a, b)


@pytest.mark.parametrize("M,K,N", [(300, 256, 384), (16, 2048, 2560), (2500, 512, 1088)])   # tile kernel, generation kernel, decode-once route
def test_nf4_double_quant_kernels_match_the_codec(M, K, N):
    """"bnb.nf4-dq": the kernels decode the 8-bit block scales themselves (one map lookup + fma per 64 weights).  HIP dequantise
    == host codec bit for bit, and the linear (forward and input gradient) == dense math on the dequantised weight."""
    from fastmax_experiments_amd import lora
    g = torch.Generator().manual_seed(K + N)
    lin = torch.nn.Linear(K, N)
    torch.nn.init.normal_(lin.weight, generator=g)
    q = lora.NF4Linear.from_linear(lin, double_quant=True)
    host = q.dequantize(torch.float32)
    q = q.cuda()
    assert q.double_quant and q.weight.quant_state[0].dtype == torch.uint8 and q.weight.quant_state[4][1][0].is_cuda
    wd = q.dequantize(torch.float32)
    assert torch.equal(wd.cpu(), host)
    assert torch.equal(q.dequantize(torch.bfloat16).cpu(), host.to(torch.bfloat16))
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda().requires_grad_(True)
    y = q(x)
    ref = x.detach().float() @ wd.to(torch.bfloat16).float().T + q.bias.float()
    assert _rel(y, ref) < 1.5e-2
    gy = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda()
    y.backward(gy)
    assert _rel(x.grad, gy.float() @ wd.to(torch.bfloat16).float()) < 1.5e-2


def test_qlora_layer_with_double_quantised_base():
    from fastmax_experiments_amd import lora
    torch.manual_seed(5)
    layer = lora.LoRAQKVLinear(256, (8 + 4) * 32, n_head=8, n_query_groups=2, r=8, lora_alpha=16, enable_lora=(True, False, True))
    torch.nn.init.normal_(layer.lora_B, std=0.05)
    layer.quantize_base(double_quant=True).cuda()
    for rows in (75, 1200):                                      # fused kernel / decode-once route (2 x 1200 rows >= DENSE_M)
        x = torch.randn(2, rows, 256, device="cuda", dtype=torch.bfloat16, requires_grad=True)
        y = layer(x)
        assert _rel(y, _dense_reference(layer, x.detach())) < 2e-2
        y.backward(torch.randn_like(y))
        assert torch.isfinite(layer.lora_A.grad).all() and layer.linear.weight.grad is None


@pytest.mark.parametrize("M,N,K,rp", [(2048, 2560, 2048, 16), (300, 320, 128, 32), (1000, 64, 448, 16), (513, 1096, 192, 0)])
@pytest.mark.parametrize("dq", [False, True])
def test_hand_written_qlora_gemm(M, N, K, rp, dq):
    """csrc/nf4_gemm.hip (256 x 256 tiles, or 128 x 256 when those would leave CUs idle; x by LDS-DMA, bias + LoRA step fused):
    dense bf16 weight and NF4 codes decoded in the loop, ragged M / N included, against float32 math on the same dequantised weight; and the transposed dequantise."""
    from fastmax_experiments_amd import lora
    g = torch.Generator().manual_seed(M + N + K)
    lin = torch.nn.Linear(K, N)
    torch.nn.init.normal_(lin.weight, std=0.05, generator=g)
    q = lora.NF4Linear.from_linear(lin, double_quant=dq).cuda()
    scales = lora.NF4Scales(q.weight.quant_state)
    wd = q.dequantize(torch.bfloat16)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    ea = eb = None
    ref = x.float() @ wd.float().T + q.bias.float()
    if rp:
        ea = (torch.randn(M, rp, generator=g) * 0.1).to(torch.bfloat16).cuda()
        eb = (torch.randn(N, rp, generator=g) * 0.1).to(torch.bfloat16).cuda()
        ref = ref + ea.float() @ eb.float().T
    from fastmax_experiments_amd import _lib
    y_dense = lora.hip_gemm(x, wd, None, q.bias, ea, eb, N)            # few tiles: the 128-row-tile kernel
    _lib.check(_lib.lib().fastmax_hip_tune(b"gemm_sched", 13), "tune")   # ... and with it forbidden: 256 x 256 tiles
    try:
        y_dense256 = lora.hip_gemm(x, wd, None, q.bias, ea, eb, N)
    finally:
        _lib.check(_lib.lib().fastmax_hip_tune(b"gemm_sched", 0), "tune")
    y_nf4 = lora.hip_gemm(x, q.weight.data, scales, q.bias, ea, eb, N)
    assert y_dense.shape == (M, N) and _rel(y_dense, ref) < 1e-2 and _rel(y_nf4, ref) < 1e-2
    assert torch.equal(y_dense, y_nf4)                      # same bf16 weight values either way -> the same bits
    assert torch.equal(y_dense, y_dense256)                 # same accumulation order per element in both tilings
    if N % 64 == 0 and K % 64 == 0:
        wt = lora._dense_weight_t(q.weight.data, scales, N, K)
        assert torch.equal(wt, wd.t())


@pytest.mark.parametrize("route", ["gemm", "fused", "library"])
def test_training_size_qlora_layer_routes_agree(route, monkeypatch):
    """LoRAQKVLinear at a training row count through the three routes of the frozen product (hand-written GEMM on the decoded
    weight -- the default --, NF4 decoded inside the GEMM loop, decode once + library GEMM): forward and all gradients against
    dense float32 math"""
    from fastmax_experiments_amd import lora
    monkeypatch.setattr(lora, "QLORA_ROUTE", route)
    torch.manual_seed(7)
    layer = lora.LoRAQKVLinear(256, (8 + 4) * 32, n_head=8, n_query_groups=2, r=8, lora_alpha=16, enable_lora=(True, False, True))
    torch.nn.init.normal_(layer.lora_B, std=0.05)
    layer.quantize_base().cuda()
    lora.mark_only_lora_as_trainable(layer)
    x = torch.randn(3, 1000, 256, device="cuda", dtype=torch.bfloat16, requires_grad=True)      # 3000 rows >= DENSE_M
    y = layer(x)
    assert _rel(y, _dense_reference(layer, x.detach())) < 2e-2
    gy = torch.randn_like(y)
    y.backward(gy)
    xa = x.detach().float().requires_grad_(True)
    A = layer.lora_A.detach().float().requires_grad_(True)
    B = layer.lora_B.detach().float().requires_grad_(True)
    wd = layer.linear.dequantize(torch.float32)
    after_A = F.linear(xa, A)
    after_B = layer.conv1d(after_A.transpose(-2, -1), B.unsqueeze(-1)).transpose(-2, -1)
    yr = xa @ wd.T + layer.linear.bias.float() + layer.zero_pad(after_B) * layer.scaling
    yr.backward(gy.float())
    assert _rel(x.grad, xa.grad) < 3e-2 and _rel(layer.lora_A.grad, A.grad) < 3e-2 and _rel(layer.lora_B.grad, B.grad) < 3e-2


def test_plain_gemm_loop_matches_the_default_kernel():
    """the A/B baseline kept from round 2 ("gemm_sched" 14: every wave issues its copies, reads its fragments, multiplies): same
    sums in the same order per element as the default 8-wave kernel -> the same bits, bias and LoRA step included"""
    from fastmax_experiments_amd import _lib, lora
    g = torch.Generator().manual_seed(14)
    M, N, K = 1024, 768, 320
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).cuda()
    bias = torch.randn(N, generator=g).cuda()
    ea = (torch.randn(M, 16, generator=g) * 0.1).to(torch.bfloat16).cuda()
    eb = (torch.randn(N, 16, generator=g) * 0.1).to(torch.bfloat16).cuda()
    L = _lib.lib()
    _lib.check(L.fastmax_hip_tune(b"gemm_sched", 13), "tune")        # the default 8-wave 256 x 256 kernel (128-row tiles forbidden)
    try:
        want = lora.hip_gemm(x, w, None, bias, ea, eb, N)
        _lib.check(L.fastmax_hip_tune(b"gemm_sched", 14), "tune")
        got = lora.hip_gemm(x, w, None, bias, ea, eb, N)
        got_plain = lora.hip_gemm(x, w, None, None, None, None, N)
    finally:
        _lib.check(L.fastmax_hip_tune(b"gemm_sched", 0), "tune")
    ref = x.float() @ w.float().T
    assert _rel(got_plain, ref) < 1e-2
    assert torch.equal(got, want)


@pytest.mark.parametrize("kind", ["linear", "qkv"])
def test_dense_base_lora_layers_on_the_tile_gemm(kind, monkeypatch):
    """LoRA on an UNQUANTISED frozen bf16 base (finetune/lora.py without --quantize; lit_gpt/lora.py:170-177, 419-433) at a
    training row count: the hand-written tile GEMM with bias + LoRA step fused (dx through W^T, dA / dB by the rank-r kernels)
    against the tensor-op form of the same layer"""
    from fastmax_experiments_amd import lora
    torch.manual_seed(11)
    if kind == "linear":
        layer = lora.LoRALinear(256, 384, r=8, lora_alpha=16, bias=True)
    else:
        layer = lora.LoRAQKVLinear(256, (8 + 4) * 32, n_head=8, n_query_groups=2, r=8, lora_alpha=16, enable_lora=(True, False, True),
                                   bias=False)
    torch.nn.init.normal_(layer.lora_B, std=0.05)
    layer = layer.to(torch.bfloat16).cuda()
    lora.mark_only_lora_as_trainable(layer)
    x = torch.randn(2, 1500, 256, device="cuda", dtype=torch.bfloat16)
    gy = None
    res = []
    for flag in ("1", "0"):
        monkeypatch.setenv("FASTMAX_DENSE_LORA_GEMM", flag)
        xa = x.clone().requires_grad_(True)
        assert layer._dense_base_on_tile_gemm(xa) == (flag == "1")
        y = layer(xa)
        gy = torch.randn_like(y) if gy is None else gy
        y.backward(gy)
        res.append((y.detach().float(), xa.grad.float(), layer.lora_A.grad.float().clone(), layer.lora_B.grad.float().clone()))
        layer.lora_A.grad = layer.lora_B.grad = None
    for got, want in zip(*res):
        assert float((got - want).abs().max()) <= 3e-2 * float(want.abs().max())


@pytest.mark.gpu
def test_dense_base_with_a_trainable_bias_keeps_its_gradient():
    """mark_only_lora_as_trainable(bias="all") (lit_gpt/lora.py:436-461): the tile GEMM's autograd function treats the base as
    frozen data, so a layer whose bias trains must stay on the tensor-op route and deliver d(bias); the transposed copy of a
    frozen dense weight is built once per layer and rebuilt when the weight is written"""
    from fastmax_experiments_amd import lora
    torch.manual_seed(5)
    layer = lora.LoRALinear(256, 384, r=8, lora_alpha=16, bias=True).to(torch.bfloat16).cuda()
    torch.nn.init.normal_(layer.lora_B, std=0.05)
    lora.mark_only_lora_as_trainable(layer, bias="all")
    x = torch.randn(2, 1500, 256, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    assert layer.linear.bias.requires_grad and not layer._dense_base_on_tile_gemm(x)
    layer(x).sum().backward()
    assert layer.linear.bias.grad is not None and float(layer.linear.bias.grad.float().abs().sum()) > 0
    # frozen bias: the tile GEMM route, with W^T cached per layer
    lora.mark_only_lora_as_trainable(layer)
    assert layer._dense_base_on_tile_gemm(x)
    for _ in range(2):
        x.grad = None
        layer(x).sum().backward()
    wt = lora._frozen_wt[layer.linear][1]
    assert torch.equal(wt, layer.linear.weight.detach().t())
    first = x.grad.clone()
    with torch.no_grad():
        layer.linear.weight.mul_(0.5)                                # written in place: the version counter moves
    x.grad = None
    layer(x).sum().backward()
    assert lora._frozen_wt[layer.linear][1] is not wt
    assert not torch.equal(x.grad, first)


@pytest.mark.gpu
def test_decoded_weights_stay_resident_across_passes(monkeypatch):
    """the training route decodes a frozen NF4 weight (and its transpose, for dx) ONCE per layer and keeps both within
    FASTMAX_DENSE_RESIDENT_BYTES: same results as the per-call decode, entries equal to the codec's values, a re-quantised
    layer starts over, and a zero budget falls back to the scratch decode"""
    from fastmax_experiments_amd import lora
    torch.manual_seed(3)
    layer = lora.LoRALinear(256, 384, r=8, lora_alpha=16, bias=True)
    torch.nn.init.normal_(layer.lora_B, std=0.05)
    layer.quantize_base().cuda().to(torch.bfloat16)
    lora.mark_only_lora_as_trainable(layer)
    x = torch.randn(2304, 256, device="cuda", dtype=torch.bfloat16)
    gy = torch.randn(2304, 384, device="cuda", dtype=torch.bfloat16)

    def run():
        xa = x.clone().requires_grad_(True)
        y = layer(xa)
        y.backward(gy)
        out = (y.detach().clone(), xa.grad.clone(), layer.lora_A.grad.clone(), layer.lora_B.grad.clone())
        layer.lora_A.grad = layer.lora_B.grad = None
        return out

    monkeypatch.setattr(lora, "RESIDENT_BYTES", 0)
    want = run()
    assert layer.linear not in lora._resident or lora._resident[layer.linear]["w"] is None
    monkeypatch.setattr(lora, "RESIDENT_BYTES", 1 << 30)
    got = run()
    ent = lora._resident[layer.linear]
    assert ent["w"] is not None and ent["wt"] is not None
    assert torch.equal(ent["w"], layer.linear.dequantize(torch.bfloat16)) and torch.equal(ent["wt"], ent["w"].t())
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    w_ptr = ent["w"].data_ptr()
    run()
    assert lora._resident[layer.linear]["w"].data_ptr() == w_ptr            # not decoded again
    with torch.no_grad():
        layer.lora_B.mul_(2.0)
    layer.merge()                                                            # new codes: the resident copies are stale
    layer.merged = False
    run()
    assert not torch.equal(lora._resident[layer.linear]["w"], ent["w"])
