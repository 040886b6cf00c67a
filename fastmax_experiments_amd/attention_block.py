"""The caller of the hot path, as a host for parity tests and the data-parallel fine-tune step.

Reproduces the CALL CONTRACT of the reference's ``CausalSelfAttention.forward`` with LoRA / QLoRA layers
(lit_gpt/model.py:380-458 with the ``attn_alg`` dispatch at 432-451, LoRA wiring lit_gpt/lora.py:565-604):
    qkv = attn(x)                      LoRAQKVLinear (4-bit NF4 base when quantised)          model.py:392
    view / permute / split / GQA expand / reshape to (B, n_head, T, head_size)               model.py:397-420
    RoPE on the first rope_n_elem dims (apply_rope, model.py:702-708)                       model.py:422-425
    y = fastmax(q,k,v,p=2,mask) | fastmax_hack(q,k,v,p=1,mask)  -- on the DEVICE tensors     model.py:460-487
    y.reshape(B, T, head_size * n_head)   (no transpose on these branches: quirk Q3)        model.py:453-455
    proj(y)                            LoRALinear                                            model.py:458
It is deliberately NOT a port of lit-gpt's GPT: no config registry, KV cache, MLP or norms -- only the
attention sub-layer that hands tensors to the operator.  Everything outside the two LoRA linears and the
attention operator is stock tensor plumbing.
"""
import torch
import torch.nn as nn

from . import ops
from .attention_mechanisms.fastmax import fastmax
from .attention_mechanisms.fastmax_hack import fastmax_hack, fastmax_hack_grouped, grouped_route_supported
from .lora import LoRALinear, LoRAQKVLinear


def build_rope_cache(seq_len: int, n_elem: int, device=None, base: int = 10000, condense_ratio: int = 1):
    """cos / sin tables of lit_gpt/model.py:676-699 (public RoPE formula)."""
    theta = 1.0 / (base ** (torch.arange(0, n_elem, 2, device=device).float() / n_elem))
    seq_idx = torch.arange(seq_len, device=device) / condense_ratio
    idx_theta = torch.outer(seq_idx, theta).repeat(1, 2)
    return torch.cos(idx_theta), torch.sin(idx_theta)


def apply_rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    half = x.size(-1) // 2
    rotated = torch.cat((-x[..., half:], x[..., :half]), dim=-1)
    return ((x * cos) + (rotated * sin)).to(dtype=x.dtype)


class CausalSelfAttention(nn.Module):
    def __init__(self, n_embd: int, n_head: int, n_query_groups: int = None, head_size: int = None, bias: bool = False,
                 rotary_percentage: float = 1.0, attn_alg: str = "fastmax", r: int = 8, alpha: int = 16,
                 dropout: float = 0.0, to_query: bool = True, to_key: bool = False, to_value: bool = True,
                 to_projection: bool = False):
        super().__init__()
        self.n_head = n_head
        self.n_query_groups = n_query_groups or n_head
        self.head_size = head_size or n_embd // n_head
        self.rope_n_elem = int(rotary_percentage * self.head_size)
        if attn_alg not in ("fastmax", "linearmax"):
            raise ValueError(f"Attention algorithm {attn_alg} not supported")          # model.py:450-451
        self.attn_alg = attn_alg
        self.fused_neighbours = True          # False: tensor-op slicing instead of the one-pass HIP kernel (A/B, parity tests)
        self.group_views = True               # grouped-query heads: K, V never copied per query head (False: the reference's expand)
        shape = (n_head + 2 * self.n_query_groups) * self.head_size
        self.attn = LoRAQKVLinear(n_embd, shape, n_head=n_head, n_query_groups=self.n_query_groups, r=r, lora_alpha=alpha,
                                  lora_dropout=dropout, enable_lora=(to_query, to_key, to_value), bias=bias)
        self.proj = LoRALinear(self.head_size * n_head, n_embd, r=(r if to_projection else 0), lora_alpha=alpha,
                               lora_dropout=dropout, bias=bias)

    def quantize_base(self, double_quant: bool = False):
        """QLoRA: both frozen linears become 4-bit NF4 (what the bnb precision plugin does in the reference);
        ``double_quant`` = the "bnb.nf4-dq" mode (finetune/lora.py:38)."""
        self.attn.quantize_base(double_quant)
        self.proj.quantize_base(double_quant)
        return self

    def forward(self, x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, input_pos=None) -> torch.Tensor:
        B, T, C = x.size()
        q_per_kv = self.n_head // self.n_query_groups
        total_qkv = q_per_kv + 2
        if self._one_kernel_qkv(x, input_pos, B, T, q_per_kv):
            return self._forward_one_kernel_qkv(x, cos, sin, B, T, q_per_kv)
        qkv = self.attn(x)
        fused = (self.fused_neighbours and x.device.type == "cuda" and input_pos is None and
                 ops.rope_qkv_supported(qkv.dtype, self.head_size, self.rope_n_elem))
        grouped = (fused and self.attn_alg == "linearmax" and q_per_kv > 1 and torch.is_grad_enabled() and
                   (x.requires_grad or any(p.requires_grad for p in self.attn.parameters())) and
                   grouped_route_supported(x.device, qkv.dtype, self.head_size, B * self.n_head))
        views = fused and q_per_kv > 1 and self.group_views
        if grouped:
            # training, grouped-query heads: K stays at its n_query_groups heads through RoPE and the linearmax prologue
            # (statistics and gradient once per key head).  group_views: neither K nor V is ever copied per query head --
            # (batch, group) is the kernels' batch axis and K, V are stride-0 views; else the prologue's store writes the copies
            q, k, v = ops.RopeQKVSplit.apply(qkv.view(B, T, self.n_query_groups, total_qkv, self.head_size), cos, sin,
                                             self.rope_n_elem, 4 if views else 2)
            y = fastmax_hack_grouped(q, k, v, q_per_kv, p=1)
            y = y.reshape(B, T, self.head_size * self.n_head)      # model.py:453-455 (no transpose: quirk Q3)
            return self.proj(y)
        elif fused:
            # de-interleave + RoPE (+ GQA expand, or group views: see ops.RopeQKVSplit) in one HIP pass
            q, k, v = ops.RopeQKVSplit.apply(qkv.view(B, T, self.n_query_groups, total_qkv, self.head_size), cos, sin,
                                             self.rope_n_elem, 3 if views else 1)
        else:
            # shapes the one-pass kernel does not take (decode with input_pos, rotary widths that are not whole 16-byte
            # pieces): plain slicing of the (B, T, group, slot, hs) view -- slots 0..q_per_kv-1 are the group's query heads,
            # then its key head, then its value head (model.py:397-420) -- with K, V repeated per query head
            qkv5 = qkv.view(B, T, self.n_query_groups, total_qkv, self.head_size)
            q = qkv5[:, :, :, :q_per_kv].permute(0, 2, 3, 1, 4).reshape(B, self.n_head, T, self.head_size)
            k, v = (qkv5[:, :, :, q_per_kv + i].permute(0, 2, 1, 3).repeat_interleave(q_per_kv, dim=1) for i in (0, 1))
            n = self.rope_n_elem
            q = torch.cat((apply_rope(q[..., :n], cos, sin), q[..., n:]), dim=-1)
            k = torch.cat((apply_rope(k[..., :n], cos, sin), k[..., n:]), dim=-1)
        mask = input_pos is None                                   # model.py:462-466, 477-481
        if self.attn_alg == "linearmax":
            y = fastmax_hack(q, k, v, p=1, mask=mask)              # model.py:472
        else:
            y = fastmax(q, k, v, p=2, mask=mask)                   # model.py:485, minus the .cpu()/.cuda() hops
        y = y.reshape(B, T, self.head_size * self.n_head)          # model.py:453-455 (no transpose: quirk Q3)
        return self.proj(y)


# (methods of CausalSelfAttention, kept below forward for readability)
def _one_kernel_qkv(self, x, input_pos, B, T, q_per_kv) -> bool:
    """can the qkv projection, the de-interleave and RoPE run as ONE kernel (nf4_gemm.hip's tile epilogue)?  Training-size bf16
    input on the hand-written GEMM route, whole heads per 256-column tile, and a K / V layout that needs no per-head copies
    (group views, or one query head per group)"""
    from . import lora
    attn = self.attn
    if not (self.fused_neighbours and self.gemm_rope and x.device.type == "cuda" and input_pos is None and x.dtype == torch.bfloat16):
        return False
    if not (isinstance(attn, lora.LoRAQKVLinear) and attn.rope_fusable(x)):
        return False
    if q_per_kv > 1 and not self.group_views:
        return False
    N, K = attn.linear.out_features, attn.linear.in_features
    return (ops.rope_qkv_supported(x.dtype, self.head_size, self.rope_n_elem) and
            lora.gemm_rope_supported(N, K, B * T, T, self.n_query_groups, q_per_kv, self.head_size, self.rope_n_elem))


def _forward_one_kernel_qkv(self, x, cos, sin, B, T, q_per_kv):
    tables16 = cos.dtype == x.dtype and x.dtype in (torch.bfloat16, torch.float16)
    cos32, sin32 = ops._rope_tables_f32(cos, sin, T, self.rope_n_elem)
    grouped = (self.attn_alg == "linearmax" and q_per_kv > 1 and torch.is_grad_enabled() and
               (x.requires_grad or any(p.requires_grad for p in self.attn.parameters())) and
               grouped_route_supported(x.device, x.dtype, self.head_size, B * self.n_head))
    expand = (4 if grouped else 3) if q_per_kv > 1 else 0
    q, k, v = self.attn(x, rope=(cos32, sin32, B, T, self.n_query_groups, q_per_kv, self.head_size, self.rope_n_elem, tables16, expand))
    if grouped:
        y = fastmax_hack_grouped(q, k, v, q_per_kv, p=1)
    elif self.attn_alg == "linearmax":
        y = fastmax_hack(q, k, v, p=1, mask=True)
    else:
        y = fastmax(q, k, v, p=2, mask=True)
    return self.proj(y.reshape(B, T, self.head_size * self.n_head))


CausalSelfAttention._one_kernel_qkv = _one_kernel_qkv
CausalSelfAttention._forward_one_kernel_qkv = _forward_one_kernel_qkv
CausalSelfAttention.gemm_rope = True


# head shapes of the BASELINE.json configs (lit_gpt/config.py:197-205, 1394-1411, 735-747) and of the reference config with the
# largest head size (pythia-1b, config.py:246-254: 8 heads of 256)
CONFIG_SHAPES = {
    "pythia-14m": dict(n_embd=128, n_head=4, n_query_groups=4, head_size=32, rotary_percentage=0.25, bias=True),
    "tiny-llama-1.1b": dict(n_embd=2048, n_head=32, n_query_groups=4, head_size=64, rotary_percentage=1.0, bias=False),
    "Llama-2-7b-hf": dict(n_embd=4096, n_head=32, n_query_groups=32, head_size=128, rotary_percentage=1.0, bias=False),
    "pythia-1b": dict(n_embd=2048, n_head=8, n_query_groups=8, head_size=256, rotary_percentage=0.25, bias=True),
    "Gemma-2b": dict(n_embd=2048, n_head=8, n_query_groups=1, head_size=256, rotary_percentage=1.0, bias=False),   # config.py:796-811 (multi-query)
}
