"""Drop-in for the reference's ``attention_mechanisms/fastmax.py`` (same names, same
signatures), backed by hand-written HIP kernels for MI355X instead of einsum + cumsum.

Reference interface mirrored here (file:line in /root/reference):
  fastmax(...)                         attention_mechanisms/fastmax.py:7-27
  fastattention_einops.forward         :41-111   (normalize_term rule 78-82, o = F/g 97,
                                                  create_attn branch 99-104, saved tensors 106-109)
  fastattention_einops.backward        :113-182  (returns dq, dk, dv + six None)
  fastattention_einops.normalize       :326-334
  fastattention_einops.compute_attn    :336-381

Behaviour kept from the reference:
  Q1  masked: output dtype == input dtype; unmasked: bf16/fp16 inputs give a float32 output
  Q2  ``dropout_rate`` is accepted and ignored; ``mask is False`` selects the unmasked branch,
      anything else is the causal branch
  Q4  unmasked N_q != N_k is accepted; the denominator's constant term is N_q
  Q5  p must be 1 or 2 -> ValueError (raised at forward time here)
  Q6  create_attn=True is forward-only (the reference saves g=None and cannot backprop)
Differences, on purpose:
  * tensors may live on the HIP device (no .cpu() hop needed); CPU tensors are accepted like in
    lit_gpt/model.py:482-486 -- they are staged to the current HIP device, computed there, and the
    result is returned on the CPU.  Without a HIP device the call raises: there is no CPU fallback.
  * float64 inputs are computed with float32 arithmetic (the kernels accumulate in fp32) and
    returned as float64.
"""
import logging

import torch

from .. import ops

_KERNEL_DTYPES = (torch.float32, torch.bfloat16, torch.float16)


MAX_HEADS_PER_LAUNCH = 65535


def fastmax(q, k, v, mask=True, normalize_term=8, tensors_normalized=False, p=1, dropout_rate=0.0,
            create_attn=False):
    """Wrapper around ``fastattention_einops`` (reference: fastmax.py:7-27)."""
    B, H = q.shape[0], q.shape[1]
    if B * H > MAX_HEADS_PER_LAUNCH and B > 1 and create_attn is False:
        # (b,h) pairs ride on a 16-bit grid dimension in some kernels: run the batch in slices (heads are independent)
        step = max(1, MAX_HEADS_PER_LAUNCH // H)
        return torch.cat([fastattention_einops.apply(q[i:i + step], k[i:i + step], v[i:i + step], mask, normalize_term,
                                                     tensors_normalized, p, dropout_rate, create_attn)
                          for i in range(0, B, step)], dim=0)
    return fastattention_einops.apply(q, k, v, mask, normalize_term, tensors_normalized, p, dropout_rate,
                                      create_attn)


def _out_dtype(in_dtype, causal):
    # Q1: the unmasked branch adds a float32 torch.ones (fastmax.py:271) which promotes bf16/fp16
    if not causal and in_dtype in (torch.bfloat16, torch.float16):
        return torch.float32
    return in_dtype


class fastattention_einops(torch.autograd.Function):
    """Factorised polynomial attention, forward and hand-derived backward, on MI355X."""

    @staticmethod
    def forward(ctx, q, k, v, mask=True, normalize_term=8, tensors_normalized=False, p=1, dropout_rate=0.0,
                create_attn=False):
        D = q.shape[-1]
        nt = ops.effective_normalize_term(D, normalize_term, tensors_normalized)
        causal = mask is not False
        if p not in (1, 2):
            raise ValueError(f"p should be 1 or 2, got p={p}")
        dev = ops._device() if q.device.type != "cuda" else q.device
        home, in_dtype = q.device, q.dtype
        if in_dtype not in _KERNEL_DTYPES and in_dtype != torch.float64:
            raise TypeError(f"fastmax: unsupported dtype {in_dtype}")
        kdt = torch.float32 if in_dtype == torch.float64 else in_dtype
        qd, kd, vd = (ops._prep(t.detach().to(kdt), dev) for t in (q, k, v))

        if create_attn is not False:
            logging.warning("compute_attn = True. Performing unfactorized computations")
            a = fastattention_einops.compute_attn(qd.to(in_dtype), kd.to(in_dtype), mask, nt, p)
            o = torch.matmul(a, vd.to(in_dtype))
            ctx.forward_only = True
            ctx.mark_non_differentiable(o, a)
            return o.to(home), a.to(home)

        # head sizes that are not a whole number of 16-byte pieces would fall onto the vector-ALU kernels (measured 25-70x
        # slower at N = 2048): zero columns change neither q.k nor the populated columns of f(q.k) v, so pad to a multiple of 8
        # (nt keeps the true D) and slice the result
        Dp = D if D % 8 == 0 else min(ops.MAX_HEAD_SIZE, (D + 7) // 8 * 8)
        if Dp != D:
            qd, kd, vd = (torch.nn.functional.pad(t, (0, Dp - D)) for t in (qd, kd, vd))
        out_dt = _out_dtype(kdt, causal)
        o, g, states = ops.forward(qd, kd, vd, p, causal, nt, g0=float(q.shape[2]), out_dtype=out_dt, keep_states=True)
        ctx.save_for_backward(qd, kd, vd, o, g)
        ctx.states = states            # the forward's sequence-split prefix states (or None): the backward reuses them
        ctx.mask, ctx.normalize_term, ctx.p = mask, nt, p
        ctx.home, ctx.in_dtype, ctx.forward_only, ctx.D = home, in_dtype, False, D
        if Dp != D:
            o = o[..., :D].contiguous()
        if in_dtype == torch.float64:
            o = o.double()
        return o.to(home)

    @staticmethod
    def backward(ctx, o_grad, *unused):
        if ctx.forward_only:
            raise RuntimeError("create_attn=True is forward-only (the reference saves g=None, fastmax.py:101-106)")
        q, k, v, o, g = ctx.saved_tensors
        causal = ctx.mask is not False
        go = o_grad.to(q.device)
        if q.shape[-1] != ctx.D:                                  # padded head size: zero gradient columns in, slice out
            go = torch.nn.functional.pad(go, (0, q.shape[-1] - ctx.D))
        if o.dtype != q.dtype:
            # unmasked bf16/fp16 (Q1): o and its gradient are float32 -> differentiate in float32
            q32, k32, v32 = (ops._prep(t.float(), q.device) for t in (q, k, v))
            dq, dk, dv = ops.backward(q32, k32, v32, o, g, ops._prep(go.float(), q.device), ctx.p, causal,
                                      ctx.normalize_term)
        else:
            dq, dk, dv = ops.backward(q, k, v, o, g, ops._prep(go.to(q.dtype), q.device), ctx.p, causal,
                                      ctx.normalize_term, states=ctx.states)
        if q.shape[-1] != ctx.D:
            dq, dk, dv = (t[..., :ctx.D].contiguous() for t in (dq, dk, dv))
        dq, dk, dv = (t.to(device=ctx.home, dtype=ctx.in_dtype) for t in (dq, dk, dv))
        return dq, dk, dv, None, None, None, None, None, None

    @staticmethod
    def normalize(q, k):
        """fastmax.py:326-334 -- mean-centre each token, divide by the max token norm of the (b,h) slab."""
        dev = ops._device() if q.device.type != "cuda" else q.device
        out = []
        for t in (q, k):
            kdt = t.dtype if t.dtype in _KERNEL_DTYPES else torch.float32
            y, _ = ops.normalize(ops._prep(t.detach().to(kdt), dev))
            out.append(y.to(device=t.device, dtype=t.dtype))
        return out[0], out[1]

    @staticmethod
    def compute_attn(q, k, mask, normalize_term, p):
        """Dense attention matrix a = f(s)*mask / rowsum (fastmax.py:336-381): the debugging /
        visualisation path, O(N^2) memory by definition, written with device tensor ops."""
        if p == 1:
            f = lambda x: 1 + x
        elif p == 2:
            f = lambda x: 1 + x + x ** 2 / 2
        else:
            raise ValueError(f"p should be 1 or 2, got p={p}")
        s = torch.matmul(q, k.transpose(-1, -2)) / normalize_term
        fs = f(s)
        if mask is not False:
            fs = torch.tril(fs)
        return fs / fs.sum(-1, keepdim=True)
