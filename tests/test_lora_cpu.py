"""Host logic of the QLoRA layers, pinned by the data the reference's own tests hold for lit_gpt/lora.py
(tests/test_lora.py:83-159 lora_ind / zero_pad / non-zero columns, :333-369 conv1d equivalence, :37-80 merge),
plus the NF4 codec's self-consistency (bitsandbytes is absent: NF4 values are 'parity unpinned')."""
import pytest
import torch
from torch.nn import functional as F

from fastmax_experiments_amd import lora


def _qkv(n_head, n_query_groups, n_embd=8, r=2, enable=(True, False, True)):
    hs = n_embd // n_head
    return lora.LoRAQKVLinear(n_embd, (n_head + 2 * n_query_groups) * hs, n_head=n_head, n_query_groups=n_query_groups,
                              r=r, lora_alpha=8, lora_dropout=0.1, enable_lora=enable)


@pytest.mark.parametrize("groups,out,ind,bshape", [
    (4, 24, [0, 1, 6, 7, 12, 13, 18, 19, 4, 5, 10, 11, 16, 17, 22, 23], (16, 2)),      # MHA
    (1, 12, [0, 1, 2, 3, 4, 5, 6, 7, 10, 11], (10, 2)),                                # MQA
    (2, 16, [0, 1, 2, 3, 8, 9, 10, 11, 6, 7, 14, 15], (12, 2)),                        # GQA
])
def test_lora_mqa_gqa_pins(groups, out, ind, bshape):
    attn = _qkv(4, groups)
    for p in attn.linear.parameters():
        torch.nn.init.zeros_(p)
    torch.nn.init.ones_(attn.lora_B)
    assert attn.linear.weight.shape == (out, 8)
    assert attn.lora_A.shape == (4, 8) and attn.lora_B.shape == bshape
    assert attn.lora_ind == ind
    x = torch.randint(0, 8, size=(3, 5, len(ind)), dtype=torch.int64)
    assert attn.zero_pad(x).shape == (3, 5, out)
    attn.eval()
    y = attn(torch.randn(2, 30, 8))
    non = list(set(range(out)).difference(ind))
    assert torch.count_nonzero(y[:, :, ind]) == 2 * 30 * len(ind)
    assert torch.count_nonzero(y[:, :, non]) == 0


@torch.inference_mode()
@pytest.mark.parametrize("n_head", (1, 2, 3, 6, 12))
@pytest.mark.parametrize("enable_lora", [(False, False, True), (False, True, False), (False, True, True),
                                         (True, False, False), (True, False, True), (True, True, False),
                                         (True, True, True)])
def test_conv1d_equivalence_and_dense_operand(n_head, enable_lora):
    C = 12
    torch.manual_seed(n_head * 8 + sum(4 >> i for i, e in enumerate(enable_lora) if e))
    layer = lora.LoRAQKVLinear(C, 3 * C, n_head=n_head, n_query_groups=n_head, r=2, enable_lora=enable_lora)
    torch.nn.init.normal_(layer.lora_B)
    x = torch.randn((1, 1, C))
    a = F.linear(x, layer.lora_A).transpose(-2, -1)
    b = layer.lora_B.data.unsqueeze(-1)
    ref = F.conv1d(a, b, groups=sum(layer.enable_lora))
    # the layer's block-diagonal product sums in a different order than the grouped convolution: float32 rounding apart
    assert torch.allclose(ref, layer.conv1d(a, b), atol=1e-6)
    layer.n_head = layer.n_query_groups + 1
    assert torch.allclose(ref, layer.conv1d(a, b), atol=1e-6)
    # the single dense operand the fused kernel consumes reproduces zero_pad(conv1d(.)) exactly
    after_A = F.linear(x, layer.lora_A)
    want = layer.zero_pad(layer.conv1d(after_A.transpose(-2, -1), b).transpose(-2, -1))
    got = after_A @ layer._dense_rows().T
    assert torch.allclose(want, got, atol=1e-6)


@pytest.mark.parametrize("groups", [4, 2, 1])
def test_dense_operand_gqa(groups):
    layer = _qkv(4, groups, n_embd=16, r=3, enable=(True, True, True) if groups == 2 else (True, False, True))
    torch.nn.init.normal_(layer.lora_B)
    x = torch.randn(2, 7, 16)
    after_A = F.linear(x, layer.lora_A)
    want = layer.zero_pad(layer.conv1d(after_A.transpose(-2, -1), layer.lora_B.unsqueeze(-1)).transpose(-2, -1))
    assert torch.allclose(want, after_A @ layer._dense_rows().T, atol=1e-6)
    assert torch.allclose(layer.get_lora_AB(), layer._dense_rows() @ layer.lora_A * layer.scaling, atol=1e-6)


def test_lora_merge_dense():
    layer = lora.LoRALinear(16, 24, r=4, lora_alpha=8)
    torch.nn.init.normal_(layer.lora_B)
    w0 = layer.linear.weight.data.clone()
    x = torch.randn(3, 16)
    y0 = layer(x)
    delta = layer.get_lora_AB()
    layer.merge()
    assert layer.merged and torch.allclose(layer.linear.weight.data, w0 + delta)
    assert torch.allclose(layer(x), y0, atol=1e-5)
    layer.merge()                                            # idempotent
    assert torch.allclose(layer.linear.weight.data, w0 + delta)


def test_only_lora_trainable_and_filter():
    m = torch.nn.Sequential(lora.LoRALinear(8, 8, r=2), lora.LoRAQKVLinear(8, 24, 4, 4, r=2, enable_lora=True))
    lora.mark_only_lora_as_trainable(m)
    names = {n for n, p in m.named_parameters() if p.requires_grad}
    assert names == {"0.lora_A", "0.lora_B", "1.lora_A", "1.lora_B"}
    assert lora.lora_filter("x.lora_A", None) and not lora.lora_filter("x.linear.weight", None)


# ---- NF4 codec ------------------------------------------------------------------------------
def test_nf4_codebook_and_packing():
    code = lora.NF4_CODE
    assert code.numel() == 16 and code[0] == -1 and code[7] == 0 and code[15] == 1 and torch.all(code[1:] > code[:-1])
    # weights that ARE codes times an absmax round-trip exactly; first element in the high nibble
    idx = torch.arange(64) % 16
    w = (code[idx] * 3.0).reshape(1, 64)
    packed, absmax = lora.nf4_quantize(w)
    assert packed.dtype == torch.uint8 and packed.numel() == 32 and absmax.tolist() == [3.0]
    assert packed[0].item() == (0 << 4) | 1 and packed[1].item() == (2 << 4) | 3
    assert torch.equal(lora.nf4_dequantize(packed, absmax, (1, 64)), w)


def test_nf4_error_bound_and_idempotence():
    g = torch.Generator().manual_seed(0)
    w = torch.randn(96, 128, generator=g)
    packed, absmax = lora.nf4_quantize(w)
    assert packed.numel() == w.numel() // 2 and absmax.numel() == w.numel() // 64
    d = lora.nf4_dequantize(packed, absmax, w.shape)
    gap = (lora.NF4_CODE[1:] - lora.NF4_CODE[:-1]).max() / 2
    err = (d - w).reshape(-1, 64).abs() / absmax[:, None]
    assert float(err.max()) <= float(gap) + 1e-6
    p2, a2 = lora.nf4_quantize(d)
    assert torch.equal(p2, packed) and torch.allclose(a2, absmax)
    with pytest.raises(ValueError):
        lora.nf4_quantize(torch.zeros(3, 5))


def test_nf4linear_looks_like_a_bnb_weight():
    lin = torch.nn.Linear(128, 64)
    q = lora.NF4Linear.from_linear(lin)
    assert q.weight.dtype == torch.uint8 and q.weight.numel() == 128 * 64 // 2 and not q.weight.requires_grad
    assert tuple(q.weight.quant_state[1]) == (64, 128)                 # utils.py:36-38 reads the shape here
    assert q.weight.quant_state[0].dtype == torch.float32
    assert (q.dequantize() - lin.weight.data).abs().max() < 0.2 * lin.weight.data.abs().max()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        q(torch.randn(2, 128))


def test_lora_only_checkpoint_and_merge_round_trip(tmp_path):
    """finetune/lora.py:341-343 (adapter-only save) and scripts/merge_lora.py:70-82 (load on top, merge, strip)."""
    torch.manual_seed(0)
    m = torch.nn.Sequential(lora.LoRALinear(16, 16, r=2, lora_alpha=4), lora.LoRAQKVLinear(16, 48, 4, 4, r=2, enable_lora=(True, False, True)))
    for mod in m:
        torch.nn.init.normal_(mod.lora_B)
    path = tmp_path / "lit_model_lora_finetuned.pth"
    lora.save_lora_checkpoint(m, path)
    saved = torch.load(str(path), weights_only=True)["model"]
    assert set(saved) == {"0.lora_A", "0.lora_B", "1.lora_A", "1.lora_B"}            # only lora_* keys
    fresh = torch.nn.Sequential(lora.LoRALinear(16, 16, r=2, lora_alpha=4), lora.LoRAQKVLinear(16, 48, 4, 4, r=2, enable_lora=(True, False, True)))
    fresh.load_state_dict({k: v for k, v in m.state_dict().items() if "lora_" not in k}, strict=False)
    lora.load_lora_checkpoint(fresh, path)
    x = torch.randn(3, 16)
    assert torch.allclose(fresh(x), m(x), atol=1e-6)
    lora.merge_lora_weights(fresh)
    sd = lora.merged_state_dict(fresh)
    assert set(sd) == {"0.weight", "0.bias", "1.weight", "1.bias"}
    plain = torch.nn.Sequential(torch.nn.Linear(16, 16), torch.nn.Linear(16, 48))
    plain.load_state_dict(sd)
    assert torch.allclose(plain(x), m(x), atol=1e-5)


def test_nf4linear_state_dict_round_trip():
    lin = torch.nn.Linear(128, 64)
    q = lora.NF4Linear.from_linear(lin)
    sd = q.state_dict()
    assert "_extra_state" in sd and sd["weight"].dtype == torch.uint8
    q2 = lora.NF4Linear(128, 64)
    q2.load_state_dict(sd)
    assert torch.equal(q2.dequantize(), q.dequantize())


def test_double_quantised_block_scales_codec():
    """"bnb.nf4-dq" (finetune/lora.py:38; QLoRA section 3): absmax - mean stored as 8-bit codes of the 256-level dynamic map,
    block 256, one fp32 scale per block.  bitsandbytes is absent: self-consistency only (parity unpinned)."""
    code2 = lora.dynamic_map_8bit()
    assert code2.numel() == 256 and torch.unique(code2).numel() == 256 and torch.equal(code2, code2.sort().values)
    assert float(code2[-1]) == 1.0 and 0.0 in code2.tolist() and float(code2[0]) < -0.99        # signed map, not symmetric
    g = torch.Generator().manual_seed(3)
    absmax = torch.rand(700, generator=g) * 0.2 + 0.01                    # ragged: 700 = 2 blocks of 256 + 188
    codes, absmax2, offset, c2 = lora.absmax_double_quantize(absmax)
    assert codes.dtype == torch.uint8 and codes.numel() == 700 and absmax2.numel() == 3 and offset.ndim == 0
    assert abs(float(offset) - float(absmax.mean())) < 1e-7
    back = lora.absmax_double_dequantize(codes, absmax2, offset, c2)
    # nearest-code property: no other map entry reproduces a scale better
    scaled = (absmax - offset) / absmax2[torch.arange(700) // 256]
    best = (scaled[:, None] - c2[None, :]).abs().min(dim=1).values
    assert torch.allclose((scaled - c2[codes.long()]).abs(), best, atol=1e-7)
    assert float(((back - absmax).abs() / absmax2[torch.arange(700) // 256]).max()) < 0.04      # coarsest map spacing, relative to the block


def test_nf4_double_quant_layer_state_and_copies():
    import copy
    lin = torch.nn.Linear(128, 64)
    q = lora.NF4Linear.from_linear(lin, double_quant=True)
    plain = lora.NF4Linear.from_linear(lin)
    qs = q.weight.quant_state
    assert q.double_quant and not plain.double_quant
    assert qs[0].dtype == torch.uint8 and qs[0].numel() == 128 * 64 // 64 and tuple(qs[1]) == (64, 128) and qs[5] == "nf4"
    assert torch.equal(q.weight.data, plain.weight.data)                  # the 4-bit codes do not change, only the scales
    w_dq, w = q.dequantize(), plain.dequantize()
    assert 0 < float((w_dq - w).abs().max()) < 0.02 * float(w.abs().max())
    # deepcopy (nn.Parameter.__deepcopy__ calls type(self)(data, requires_grad)) keeps an independent quant_state
    q2 = copy.deepcopy(q)
    assert q2.weight.quant_state is not qs and q2.weight.quant_state[0] is not qs[0] and torch.equal(q2.dequantize(), w_dq)
    p2 = copy.deepcopy(plain.weight)
    assert isinstance(p2, lora.Params4bit) and torch.equal(p2.quant_state[0], plain.weight.quant_state[0])
    # state_dict round trip carries the double-quantised scales
    q3 = lora.NF4Linear(128, 64)
    q3.load_state_dict(q.state_dict())
    assert q3.double_quant and torch.equal(q3.dequantize(), w_dq)
    # merge keeps the mode (lora.py:142-168: dequantise + add + requantise)
    lay = lora.LoRALinear(128, 64, r=4, lora_alpha=8)
    torch.nn.init.normal_(lay.lora_B, std=0.05)
    lay.quantize_base(double_quant=True)
    before = lay.linear.dequantize()
    lay.merge()
    assert lay.linear.double_quant and float((lay.linear.dequantize() - before - lay.get_lora_AB()).abs().max()) < 0.2 * float(before.abs().max())


def test_mark_only_lora_as_trainable_bias_modes():
    """lit_gpt/lora.py:436-461: "none" freezes every bias, "all" thaws every parameter named *bias*, "lora_only" thaws the
    `bias` attribute of LoRA layers (LoRALinear keeps its bias inside `.linear`, so -- as in the reference -- nothing thaws),
    anything else raises NotImplementedError"""
    from fastmax_experiments_amd import lora
    m = torch.nn.Sequential(lora.LoRALinear(8, 8, r=2, bias=True), torch.nn.Linear(8, 4, bias=True))
    flags = lambda: {n: p.requires_grad for n, p in m.named_parameters()}
    lora.mark_only_lora_as_trainable(m)
    assert flags() == {"0.lora_A": True, "0.lora_B": True, "0.linear.weight": False, "0.linear.bias": False, "1.weight": False,
                       "1.bias": False}
    lora.mark_only_lora_as_trainable(m, bias="all")
    assert flags()["0.linear.bias"] and flags()["1.bias"] and not flags()["1.weight"] and flags()["0.lora_A"]
    lora.mark_only_lora_as_trainable(m, bias="lora_only")
    assert not flags()["0.linear.bias"] and not flags()["1.bias"]
    with pytest.raises(NotImplementedError):
        lora.mark_only_lora_as_trainable(m, bias="some")
