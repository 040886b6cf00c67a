"""A launchable data-parallel QLoRA fine-tune step around the fastmax operator (`bench.py --workload dp_step`).

Step structure of the reference's `finetune/lora.py:fit` (207-226) with the batch sharded over one process per GPU:
    for each of `gradient_accumulation_iters` micro-batches:   logits -> loss -> backward(loss / iters)     (no collective)
    at the boundary:  ONE all-reduce of the flat LoRA-gradient bucket (RCCL over xGMI), AdamW step, zero_grad
(`lit_gpt/args.py:40-57` for the accumulation arithmetic; the reference itself runs FSDP through Lightning Fabric and refuses
quantisation with devices > 1, finetune/lora.py:80-85 -- there is no reference behaviour to match beyond "same update as one
process with the same global batch", which tests/test_dp_gloo.py checks).

The model is NOT lit-gpt's GPT: it is `n_layer` attention sub-layers (NF4 base + LoRA on q, v; fastmax or linearmax) with
residual connections and a frozen lm-head feeding the chunked cross entropy -- the parts of the step the hot path owns.
Inputs are synthetic hidden states (no embedding table, no tokenizer, no checkpoint: none can be fetched here).
"""
from __future__ import annotations

import time
from typing import Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from . import dp


class AttentionStack(nn.Module):
    """n_layer x (x + CausalSelfAttention(x)) with QLoRA linears, then the lm-head loss of finetune/lora.py:216-219."""

    def __init__(self, config: str, n_layer: int, attn_alg: str, vocab: int = 32000, r: int = 8, alpha: int = 16,
                 lora_dropout: float = 0.05):
        """r, alpha, lora_dropout: the reference's fine-tune defaults (finetune/lora.py:40-42)"""
        super().__init__()
        from .attention_block import CONFIG_SHAPES, CausalSelfAttention
        shape = CONFIG_SHAPES[config]
        self.n_embd = shape["n_embd"]
        self.blocks = nn.ModuleList(CausalSelfAttention(attn_alg=attn_alg, r=r, alpha=alpha, dropout=lora_dropout, **shape) for _ in range(n_layer))
        for b in self.blocks:
            nn.init.normal_(b.attn.lora_B, std=0.02)              # a non-zero branch, so every LoRA gradient is exercised
        g = torch.Generator().manual_seed(1234)
        self.lm_head = nn.Parameter(torch.randn(vocab, self.n_embd, generator=g) * 0.02, requires_grad=False)
        self.rope_n_elem = self.blocks[0].rope_n_elem

    def prepare(self, device, quantize: bool = True):
        if quantize:
            for b in self.blocks:
                b.quantize_base()
        self.to(device)
        self.lm_head.data = self.lm_head.data.to(torch.bfloat16)
        return self

    def forward(self, x, cos, sin):
        for b in self.blocks:
            x = x + b(x, cos, sin)
        return x

    def loss(self, x, targets, cos, sin):
        from .loss import lm_head_cross_entropy
        return lm_head_cross_entropy(self.forward(x, cos, sin), self.lm_head, targets).float()


class ToyLoRA(nn.Module):
    """CPU stand-in with `lora_` parameters for the rehearsal of the multi-rank control flow (gloo): the attention operator
    has no CPU path, the step structure around it does not care."""

    def __init__(self, width: int = 32):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        self.base = nn.Parameter(torch.randn(width, width, generator=g) * 0.1)
        self.lora_A = nn.Parameter(torch.randn(4, width, generator=g) * 0.1)
        self.lora_B = nn.Parameter(torch.randn(width, 4, generator=g) * 0.1)

    def loss(self, x, targets, cos=None, sin=None):
        y = x @ self.base.T + (x @ self.lora_A.T) @ self.lora_B.T
        return ((y - targets) ** 2).mean()


def run(config: str, n_layer: int, attn_alg: str, seq: int, micro_batch: int, accum: int, steps: int, warmup: int, device,
        rank: int = 0, world: int = 1, toy: bool = False, precondition_ms: float = 0.0, graph: bool = False,
        lora_dropout: float = 0.05) -> dict:
    """`warmup` untimed + `steps` timed optimizer steps; -> timings (seconds / milliseconds, this rank)."""
    on_gpu = device.type == "cuda"
    multi = dist.is_available() and dist.is_initialized() and world > 1
    gen = torch.Generator(device=device).manual_seed(100 + rank)
    if toy:
        model = ToyLoRA().to(device)
        x = torch.randn(accum, micro_batch, seq, 32, device=device, generator=gen)
        tgt = torch.randn(accum, micro_batch, seq, 32, device=device, generator=gen)
        cos = sin = None
    else:
        from .attention_block import build_rope_cache
        model = AttentionStack(config, n_layer, attn_alg, lora_dropout=lora_dropout).prepare(device)
        model.train()
        x = torch.randn(accum, micro_batch, seq, model.n_embd, device=device, generator=gen).to(torch.bfloat16)
        tgt = torch.randint(0, model.lm_head.shape[0], (accum, micro_batch, seq), device=device, generator=gen)
        cos, sin = (t.to(torch.bfloat16) for t in build_rope_cache(seq, model.rope_n_elem, device=device))
    params = dp.trainable_lora_parameters(model)
    opt = torch.optim.AdamW(params, lr=1e-4)
    train = dp.TrainArgs(global_batch_size=micro_batch * accum * world, micro_batch_size=micro_batch)
    st = dp.DataParallelStepper(model, opt, train, lambda m, b: m.loss(b[0], b[1], cos, sin), time_comm=True)
    assert st.accum == accum
    if graph and on_gpu and not toy:
        st.capture((x[0], tgt[0]))

    def sync():
        if on_gpu:
            torch.cuda.synchronize(device)

    def one_step():
        loss = None
        for m in range(accum):
            loss = st.micro_step((x[m], tgt[m]))
        return loss

    def protocol():
        for _ in range(warmup):
            one_step()
        sync()
        if multi:
            dist.barrier()
        sync()
        st.comm_ms.clear()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = one_step()
        sync()
        if multi:
            dist.barrier()
        sync()
        el = time.perf_counter() - t0
        if multi:
            t = torch.tensor([el], dtype=torch.float64, device=device if dist.get_backend() != "gloo" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, float(loss)

    if precondition_ms > 0:
        t_end = time.perf_counter() + precondition_ms * 1e-3
        while time.perf_counter() < t_end:
            one_step()
            sync()
    elapsed, last_loss = protocol()
    comm = st.comm_times_ms()
    return {"elapsed_s": elapsed, "step_ms": elapsed * 1e3 / steps, "allreduce_ms": (sum(comm) / len(comm)) if comm else 0.0,
            "bucket_bytes": st.bucket.nbytes, "trainable_params": sum(p.numel() for p in params), "last_loss": last_loss,
            "tokens_per_step": micro_batch * accum * seq * world, "optimizer_steps": st.step_count}
