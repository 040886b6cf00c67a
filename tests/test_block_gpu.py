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
    if blk.attn_alg == "linearmax":
        y = orc.linearmax_fwd(qn, kn, vn, chunk=64)
    else:
        y, _ = c_oracle.fwd(qn, kn, vn, mask=True, p=2)
    y = torch.from_numpy(np.asarray(y, dtype=np.float32)).to(x.device).reshape(B, T, blk.head_size * blk.n_head)
    return lin(blk.proj, y)


@pytest.mark.parametrize("name,T,alg,quant", [
    ("pythia-14m", 1024, "fastmax", False),         # BASELINE config 2 head shape: bf16 forward, seq 1024
    ("pythia-14m", 1024, "linearmax", False),
    ("tiny-llama-1.1b", 2048, "fastmax", True),     # config 3: QLoRA (NF4) + fastmax, seq 2048
    ("Llama-2-7b-hf", 1024, "fastmax", True),       # config 4 head shape (D=128), shortened sequence
    ("Llama-2-7b-hf", 4096, "linearmax", True),     # config 5 head shape, linearmax (16k runs in bench_shapes)
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
    assert rel_err(y.float().cpu().numpy(), ref.cpu().numpy()) < 4e-2


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
