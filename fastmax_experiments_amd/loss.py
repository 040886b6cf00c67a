"""Loss path of the fine-tune step, mirroring the reference's interface:

* ``chunked_cross_entropy(logits, targets, chunk_size=128, ignore_index=-1)`` -- lit_gpt/utils.py:228-272 (accepts the list of
  lm-head chunks that ``GPT.forward(..., lm_head_chunk_size=128)`` returns, lora.py:547-550, or one tensor); the row-wise
  cross entropy runs in libfastmax_hip.so (fastmax_ce.hip): one read of the logits forward, one read + one write backward.
* ``lm_head_cross_entropy(x, weight, targets)`` -- the head and the loss together: logits = x W^T (a plain library GEMM), the
  HIP row kernel, and either the logits kept once for the backward pass (the row kernel turns them into d(logits) in place,
  dx = d(logits) W is the second plain product) or alive one row chunk at a time in both directions.

There is no CPU fallback: device tensors only.
"""
import ctypes
import os
from typing import List, Union

import torch

from . import _lib
from .ops import _DT, _stream


def _rows_forward(logits2d, targets1d, ignore_index):
    M, V = logits2d.shape
    loss = torch.empty(M, dtype=torch.float32, device=logits2d.device)
    lse = torch.empty(M, dtype=torch.float32, device=logits2d.device)
    with torch.cuda.device(logits2d.device):
        rc = _lib.lib().fastmax_hip_cross_entropy_forward(logits2d.data_ptr(), logits2d.stride(0), targets1d.data_ptr(),
                                                          loss.data_ptr(), lse.data_ptr(), M, V, ignore_index, _DT[logits2d.dtype],
                                                          _stream(logits2d.device))
    _lib.check(rc, "fastmax_hip_cross_entropy_forward")
    return loss, lse


def _rows_backward(logits2d, targets1d, lse, grad_loss, grad_scale, ignore_index, out):
    M, V = logits2d.shape
    with torch.cuda.device(logits2d.device):
        rc = _lib.lib().fastmax_hip_cross_entropy_backward(logits2d.data_ptr(), logits2d.stride(0), targets1d.data_ptr(),
                                                           lse.data_ptr(), None if grad_loss is None else grad_loss.data_ptr(),
                                                           ctypes.c_float(grad_scale), out.data_ptr(), out.stride(0), M, V,
                                                           ignore_index, _DT[logits2d.dtype], _stream(logits2d.device))
    _lib.check(rc, "fastmax_hip_cross_entropy_backward")
    return out


def _prep(logits, targets):
    if logits.device.type != "cuda":
        raise RuntimeError("the cross-entropy kernel runs on an MI355X only; there is no CPU fallback")
    logits2d = logits.reshape(-1, logits.size(-1))
    if logits2d.dtype not in _DT:
        logits2d = logits2d.float()
    if logits2d.stride(1) != 1:
        logits2d = logits2d.contiguous()
    return logits2d, targets.reshape(-1).to(torch.int64).contiguous()


class _CrossEntropyRows(torch.autograd.Function):
    """torch.nn.functional.cross_entropy(logits, targets, ignore_index=..., reduction="none") for (M, V) logits."""

    @staticmethod
    def forward(ctx, logits2d, targets1d, ignore_index):
        loss, lse = _rows_forward(logits2d, targets1d, ignore_index)
        ctx.save_for_backward(logits2d, targets1d, lse)
        ctx.ignore_index = ignore_index
        return loss.to(logits2d.dtype)

    @staticmethod
    def backward(ctx, grad_loss):
        logits2d, targets1d, lse = ctx.saved_tensors
        out = torch.empty_like(logits2d)
        _rows_backward(logits2d, targets1d, lse, grad_loss.float().contiguous(), 1.0, ctx.ignore_index, out)
        return out, None, None


def scored_rows(targets: torch.Tensor, vocab: int, ignore_index: int) -> torch.Tensor:
    """the rows the kernel scores: a label that is not `ignore_index` and lies inside [0, vocab).  torch's cross_entropy (what
    lit_gpt/utils.py:228-272 calls) raises a device assert for any other label; the kernel gives such a row loss 0 and a zero
    gradient, so the mean's denominator counts exactly these rows -- and `check_targets` raises like the reference."""
    return (targets != ignore_index) & (targets >= 0) & (targets < vocab)


def check_targets(targets: torch.Tensor, vocab: int, ignore_index: int = -1) -> None:
    """raise for labels outside [0, vocab) other than `ignore_index` (one host sync: for tests and debugging, not the step)"""
    bad = (targets != ignore_index) & ((targets < 0) | (targets >= vocab))
    if bool(bad.any()):
        raise ValueError(f"cross entropy target {int(targets[bad][0])} is outside [0, {vocab}) and is not ignore_index={ignore_index}")


def cross_entropy_rows(logits, targets, ignore_index=-100):
    logits2d, targets1d = _prep(logits, targets)
    return _CrossEntropyRows.apply(logits2d, targets1d, ignore_index)


def chunked_cross_entropy(logits: Union[torch.Tensor, List[torch.Tensor]], targets: torch.Tensor, chunk_size: int = 128,
                          ignore_index: int = -1) -> torch.Tensor:
    """lit_gpt/utils.py:228-272.  The reference chunks to bound autograd's memory spike; the kernel keeps no per-chunk
    temporaries, so every branch reduces to: per-row losses, summed, over the count of non-ignored targets (the mean that
    ``cross_entropy``'s default reduction takes when ``chunk_size == 0``)."""
    if isinstance(logits, list):
        parts = [cross_entropy_rows(chunk, t, ignore_index)
                 for chunk, t in zip(logits, targets.split(logits[0].size(1), dim=1))]
        rows = torch.cat(parts)
        vocab = logits[0].size(-1)
    else:
        rows = cross_entropy_rows(logits, targets, ignore_index)
        vocab = logits.size(-1)
    non_masked_elems = scored_rows(targets, vocab, ignore_index).sum()
    if chunk_size == 0:
        # the reference hands these branches to cross_entropy's own mean (utils.py:245, 262): NaN when no target is scored
        return rows.sum() / non_masked_elems
    return rows.sum() / non_masked_elems.clamp(min=1)                       # utils.py:256, 272: max(1, non_masked_elems)


def _kept_backward(ctx, grad_out):
    """backward pass over kept bf16 / fp32 logits: the row kernel turns them into d(logits) in place, then two plain products"""
    if getattr(ctx, "logits_consumed", False):
        raise RuntimeError("second backward pass through lm_head_cross_entropy: the kept logits were turned into d(logits) in "
                           "place by the first one (FASTMAX_HEAD_KEEP_BYTES=0 recomputes them instead)")
    x2d, weight, targets1d, lse, n, logits = ctx.saved_tensors
    # d(loss)/d(row loss) stays on the device (a per-row vector for the kernel): no host synchronisation in the step
    row_grad = (grad_out.float() / n).reshape(1).expand(logits.shape[0]).contiguous()
    _rows_backward(logits, targets1d, lse, row_grad, 1.0, ctx.ignore_index, logits)
    ctx.logits_consumed = True
    dx = logits @ weight if ctx.needs_input_grad[0] else None
    dw = (logits.t() @ x2d) if ctx.needs_input_grad[1] else None
    return dx, dw, None, None, None, None


class _LMHeadLoss(torch.autograd.Function):
    """mean cross entropy of (x W^T) against targets: library GEMM -> row kernel.  keep = False: logits alive one row chunk at
    a time, recomputed in the backward pass (nothing of size tokens x vocabulary survives the forward pass); keep = True: the
    logits of all rows are kept for the backward pass, which then needs no second x W^T product."""

    @staticmethod
    def forward(ctx, x2d, weight, targets1d, ignore_index, chunk_rows, keep=False):
        M = x2d.shape[0]
        n = scored_rows(targets1d, weight.shape[0], ignore_index).sum().clamp(min=1)
        ctx.ignore_index, ctx.chunk_rows, ctx.keep = ignore_index, chunk_rows, keep
        if keep:
            logits = x2d @ weight.t()
            loss, lse = _rows_forward(logits, targets1d, ignore_index)
            ctx.save_for_backward(x2d, weight, targets1d, lse, n, logits)
            return (loss.sum() / n).to(x2d.dtype)
        lse = torch.empty(M, dtype=torch.float32, device=x2d.device)
        total = torch.zeros((), dtype=torch.float32, device=x2d.device)
        for r0 in range(0, M, chunk_rows):
            logits = x2d[r0:r0 + chunk_rows] @ weight.t()                  # library GEMM; freed at the end of the iteration
            loss, l = _rows_forward(logits, targets1d[r0:r0 + chunk_rows], ignore_index)
            lse[r0:r0 + chunk_rows] = l
            total += loss.sum()
        ctx.save_for_backward(x2d, weight, targets1d, lse, n)
        return (total / n).to(x2d.dtype)

    @staticmethod
    def backward(ctx, grad_out):
        if ctx.keep:
            return _kept_backward(ctx, grad_out)
        x2d, weight, targets1d, lse, n = ctx.saved_tensors
        M = x2d.shape[0]
        row_grad = (grad_out.float() / n).reshape(1).expand(M).contiguous()  # stays on the device: no host synchronisation
        dx = torch.empty_like(x2d) if ctx.needs_input_grad[0] else None
        dw = torch.zeros_like(weight, dtype=torch.float32) if ctx.needs_input_grad[1] else None
        for r0 in range(0, M, ctx.chunk_rows):
            xs = x2d[r0:r0 + ctx.chunk_rows]
            logits = xs @ weight.t()
            _rows_backward(logits, targets1d[r0:r0 + ctx.chunk_rows], lse[r0:r0 + ctx.chunk_rows], row_grad[r0:r0 + ctx.chunk_rows],
                           1.0, ctx.ignore_index, logits)                    # in place: logits become d(logits)
            if dx is not None:
                dx[r0:r0 + ctx.chunk_rows] = logits @ weight
            if dw is not None:
                dw.addmm_(logits.t().float(), xs.float())
        return dx, None if dw is None else dw.to(weight.dtype), None, None, None, None


# The head's two big products (logits = x W^T and dx = d(logits) W; the head is frozen in LoRA fine-tuning, so there is no dW) are
# PLAIN matrix products and go to the vendor library by decision; what is hand-written on this path is the row-wise cross entropy
# (online max / sum over 16-byte loads, in place d(logits)).  Rounds 1-2 also carried a hand-written head with the logits reduced
# in the GEMM's accumulators (never written): 4-17 % slower than library GEMM + row kernel at (16384, 2048 -> 32000) because
# the 256 x 256 loop runs at ~0.9 of hipBLASLt's rate (profiles/r02_head_loss.md); it was retired in round 3 instead of being
# carried as a second route.  How many bytes of logits the forward pass may keep for the backward pass (0: never keep --
# recompute; the MI355X has 288 GB):
HEAD_KEEP_BYTES = int(os.environ.get("FASTMAX_HEAD_KEEP_BYTES", str(8 << 30)))


def lm_head_cross_entropy(x: torch.Tensor, weight: torch.Tensor, targets: torch.Tensor, ignore_index: int = -1,
                          chunk_rows: int = 4096) -> torch.Tensor:
    """mean over non-ignored targets of cross_entropy(x @ weight.T, targets): the reference's
    ``model(input_ids, lm_head_chunk_size=128)`` + ``chunked_cross_entropy`` pair (finetune/lora.py:216-219) for a bias-free
    head.  When a backward pass will follow and the bf16 logits fit HEAD_KEEP_BYTES they are kept (written once by the kernel
    that reduces them); otherwise they are alive one ``chunk_rows`` block at a time in both directions."""
    if x.device.type != "cuda":
        raise RuntimeError("lm_head_cross_entropy runs on an MI355X only; there is no CPU fallback")
    x2d = x.reshape(-1, x.size(-1))
    if x2d.dtype not in _DT:
        x2d = x2d.float()
    x2d, weight, t1d = x2d.contiguous(), weight.to(x2d.dtype), targets.reshape(-1).to(torch.int64).contiguous()
    fn = _LMHeadLoss
    wants_grad = torch.is_grad_enabled() and (x2d.requires_grad or weight.requires_grad)
    keep = wants_grad and x2d.shape[0] * weight.shape[0] * x2d.element_size() <= HEAD_KEEP_BYTES
    return fn.apply(x2d, weight, t1d, ignore_index, chunk_rows, keep)
