"""Drop-in for the reference's ``attention_mechanisms/fastmax_hack.py`` ("linearmax").

Reference: fastmax_hack.py:5-60.  Masked branch (36-60): mean-centre / max-norm Q and K, then
first- or second-order fastmax with normalize_term = 1.  Unmasked branch (6-33): same prologue,
first order only (``p`` is ignored there), denominator constant N_k (line 21), float32 ``ones``
promote low-precision inputs to a float32 result.
The prologue and the attention both run in libfastmax_hip.so; gradients flow through
``_NormalizeQK`` (the reference gets them from plain autograd over its einsum graph).
"""
import os

import torch

from .. import ops
from .fastmax import MAX_HEADS_PER_LAUNCH, _KERNEL_DTYPES, fastattention_einops


class _NormalizeQK(torch.autograd.Function):
    """y = (x - mean_D x) / max_n ||x_n - mean_D x_n||   (fastmax_hack.py:38-43), in x's dtype; forward and backward
    in libfastmax_hip.so (fastmax_normalize.hip).  Head sizes that are not a whole number of 16-byte pieces take the
    float32 kernel and the tensor-op backward below."""

    @staticmethod
    def forward(ctx, x, rep=1, view=False):
        """rep > 1 (x holds the G key heads of grouped-query attention): the result serves the G * rep query heads -- as
        copies (B, G*rep, N, D) written by the prologue's store, or with ``view`` as a stride-0 view (B*G, rep, N, D) of
        the G normalised heads (nothing is copied; the attention kernels' strides do the group indexing).  Either way the
        gradient arrives per query head and the backward pass sums a group while it reads."""
        r = ops.normalize_cast(x, 1 if view else rep)
        ctx.rep = rep
        if r is not None:
            y, inv = r
            ctx.save_for_backward(x, inv)
            ctx.fused = True
            if view and rep > 1:
                B, G, N, D = x.shape
                return y.view(B * G, 1, N, D).expand(B * G, rep, N, D)
            return y
        if rep != 1:
            raise NotImplementedError("grouped normalisation needs a head size that is a whole number of 16-byte pieces")
        y, inv = ops.normalize(x)
        ctx.save_for_backward(y, inv)
        ctx.in_dtype = x.dtype
        ctx.fused = False
        return y.to(x.dtype)

    @staticmethod
    def backward(ctx, gy):
        if ctx.fused:
            x, inv = ctx.saved_tensors
            return ops.normalize_backward(x, gy, inv, ctx.rep), None, None
        y, inv = ctx.saved_tensors                     # y = xc / M, inv = 1 / M
        gy = gy.float()
        gxc = gy * inv[..., None, None]
        # M = max_n ||xc_n|| is attained at token n*: dM/dxc_{n*} = xc_{n*}/M = y_{n*};  dL/dM = -sum(gy*y)/M
        dLdM = -(gy * y).sum(dim=(2, 3)) * inv                                    # (B,H)
        nstar = (y * y).sum(-1).argmax(dim=2)                                      # (B,H)
        ystar = torch.gather(y, 2, nstar[..., None, None].expand(-1, -1, 1, y.shape[-1]))
        gxc.scatter_add_(2, nstar[..., None, None].expand(-1, -1, 1, y.shape[-1]), dLdM[..., None, None] * ystar)
        gx = gxc - gxc.mean(-1, keepdim=True)
        return gx.to(ctx.in_dtype), None, None


FUSED_TRAINING = os.environ.get("FASTMAX_LINEARMAX_FUSED_TRAIN", "1") != "0"
# the prologue's backward inside the scan kernels: 0 off, 1 the k side (dK/dV kernel), 3 both sides (default)
FUSED_PROLOGUE_BWD = int(os.environ.get("FASTMAX_LINEARMAX_FUSED_PROLOGUE_BWD", "3"))


class _LinearmaxP1(torch.autograd.Function):
    """Masked first-order linearmax (fastmax_hack.py:36-60) as ONE autograd node on the raw q, k, v: the prologue is applied by
    the scan kernels while they stage their tiles, forwards and backwards, so no normalised copy of q or k is written or kept
    (the two normalize_cast passes and the separate statistics passes of the two-node route disappear); its gradient is the
    one-pass prologue backward on the scans' dq, dk.  ``rep`` > 1 (grouped-query heads): k holds the G key heads, q and v are
    (B*G, rep, N, D) (v a stride-0 group view): K is viewed the same way, every query head of a group computes the group's
    statistics from the same rows, and the prologue backward sums the group's gradients while it reads them."""

    @staticmethod
    def forward(ctx, q, k, v, rep):
        q, k, v = (ops._prep(t, t.device) for t in (q, k, v))
        kv = k
        if rep > 1:
            B, G, N, D = k.shape
            kv = k.view(B * G, 1, N, D).expand(B * G, rep, N, D)
        r = ops.linearmax_forward_fused(q, kv, v, train=True)
        if r is None:
            raise NotImplementedError("fused linearmax training route does not cover this problem")
        o, g, inv_q, inv_k, states, nstar = r
        ctx.save_for_backward(q, k, v, o, g, inv_q, inv_k)
        ctx.states, ctx.rep = states, rep
        ctx.nstar = nstar if FUSED_PROLOGUE_BWD else None
        return o

    @staticmethod
    def backward(ctx, go):
        q, k, v, o, g, inv_q, inv_k = ctx.saved_tensors
        rep = ctx.rep
        kv = k
        if rep > 1:
            B, G, N, D = k.shape
            kv = k.view(B * G, 1, N, D).expand(B * G, rep, N, D)
        go = ops._prep(go.to(q.dtype), q.device)
        # the scan kernels apply the prologue's backward to their own tiles: both sides, or with grouped-query heads the q side
        # only (the group's dk' are summed by the prologue's backward pass below)
        fuse = 0 if ctx.nstar is None else (FUSED_PROLOGUE_BWD if rep == 1 else (FUSED_PROLOGUE_BWD & 2))
        dqn, dkn, dv = ops.linearmax_backward(q, kv, v, o, g, go, inv_q, inv_k, ctx.states, nstar=ctx.nstar, fuse=fuse)
        ctx.states = ctx.nstar = None
        dq = dqn if fuse & 2 else ops.normalize_backward(q, dqn, inv_q.view(q.shape[0], q.shape[1]), 1)
        if fuse & 1:
            dk = dkn
        elif rep > 1:
            B, G, N, D = k.shape
            inv_g = inv_k.view(B * G, rep)[:, 0].contiguous().view(B, G)
            dk = ops.normalize_backward(k, dkn.view(B, G * rep, N, D), inv_g, rep)
        else:
            dk = ops.normalize_backward(k, dkn, inv_k.view(k.shape[0], k.shape[1]), 1)
        return dq, dk, dv, None


def _fused_training_ok(q, k, v, rep=1):
    """would _LinearmaxP1 serve these tensors?  (device tensors of one kernel dtype, whole 16-byte pieces per row)"""
    if not FUSED_TRAINING or q.dtype not in _KERNEL_DTYPES or q.device.type != "cuda":
        return False
    D = q.shape[-1]
    if (D * q.element_size()) % 16 or D > 128 or q.shape[0] * q.shape[1] > MAX_HEADS_PER_LAUNCH:
        return False
    prob = ops._problem(q, q, q.dtype, q.dtype, 1, True, 1.0, 0.0)
    import ctypes
    from .. import _lib
    return bool(_lib.lib().fastmax_hip_linearmax_train_supported(ctypes.byref(prob)))


def fastmax_hack(q, k, v, p=1, mask=True):
    """linearmax (reference: fastmax_hack.py:5)."""
    if q.shape[0] * q.shape[1] > MAX_HEADS_PER_LAUNCH and q.shape[0] > 1:
        step = max(1, MAX_HEADS_PER_LAUNCH // q.shape[1])          # heads are independent: run the batch in slices
        return torch.cat([fastmax_hack(q[i:i + step], k[i:i + step], v[i:i + step], p=p, mask=mask)
                          for i in range(0, q.shape[0], step)], dim=0)
    dev = ops._device() if q.device.type != "cuda" else q.device
    home, in_dtype = q.device, q.dtype
    kdt = in_dtype if in_dtype in _KERNEL_DTYPES else torch.float32
    qd, kd = (ops._prep(t.to(kdt), dev) for t in (q, k))
    needs_grad = torch.is_grad_enabled() and any(t.requires_grad for t in (q, k, v))
    if mask and p == 1 and not needs_grad:
        # inference / forward-only: prologue fused into the matrix-core kernel, one pass over Q, K, V
        o = ops.linearmax_forward_fused(qd, kd, ops._prep(v.to(kdt), dev))
        if o is not None:
            return o.to(device=home, dtype=in_dtype)
    # training (or shapes the fused kernel does not cover): prologue and attention as separate autograd nodes,
    # both in libfastmax_hip.so; 16-bit inputs keep their dtype between the two, like the reference
    vd = ops._prep(v.to(kdt), dev)
    if mask and p == 1 and needs_grad and _fused_training_ok(qd, kd, vd):
        return _LinearmaxP1.apply(qd, kd, vd, 1).to(device=home, dtype=in_dtype)
    qn, kn = _NormalizeQK.apply(qd, 1), _NormalizeQK.apply(kd, 1)
    if not mask:
        # fastmax_hack.py:6-33: first order whatever p is; constant term N_k; result float32 for
        # low-precision inputs (float32 ones at line 21), float64 stays float64
        o = _unmasked_first_order(qn, kn, vd, float(k.shape[2]))
        out_dt = torch.float64 if in_dtype == torch.float64 else torch.float32
    else:
        o = fastattention_einops.apply(qn, kn, vd, True, 1, True, p, 0.0, False)
        out_dt = in_dtype
    return o.to(device=home, dtype=out_dt)


def fastmax_hack_grouped(q, k_groups, v, rep, p=1):
    """Masked linearmax for grouped-query attention on the training route: k_groups (B,G,N,D) with H = G * rep query heads.
    Same values as fastmax_hack(q, expand(k_groups), v, p, mask=True): the max-norm statistics of identical head copies
    are identical, so the prologue runs once per key head (the reference expands first, model.py:404-411, and normalises
    every copy, fastmax_hack.py:38-43).  Two layouts:
      q, v (B,H,N,D) with v already repeated         -> the prologue's store writes the H normalised copies of K
      q, v (B*G, rep, N, D) (v a stride-0 group view) -> K is normalised at its G heads and handed on as a stride-0 view too:
                                                        no per-query-head copy of K or V exists anywhere"""
    views = q.shape[0] == k_groups.shape[0] * k_groups.shape[1] and q.shape[1] == rep and rep > 1
    if views and p == 1 and _fused_training_ok(q, k_groups, v, rep):
        return _LinearmaxP1.apply(q, k_groups, v, rep).to(q.dtype)
    qn = _NormalizeQK.apply(q, 1)
    kn = _NormalizeQK.apply(k_groups, rep, views)
    o = fastattention_einops.apply(qn, kn, v, True, 1, True, p, 0.0, False)
    return o.to(q.dtype)


def grouped_route_supported(device, dtype, head_size, heads) -> bool:
    """can fastmax_hack_grouped serve (B * H = heads) heads of this dtype / head size?  (FASTMAX_GROUPED_K=0 turns it off)"""
    return (os.environ.get("FASTMAX_GROUPED_K", "1") != "0" and device.type == "cuda" and
            dtype in (torch.bfloat16, torch.float16, torch.float32) and
            (head_size * torch.empty((), dtype=dtype).element_size()) % 16 == 0 and heads <= MAX_HEADS_PER_LAUNCH)


class _UnmaskedNk(torch.autograd.Function):
    """Unmasked p=1 fastmax with nt=1 and the denominator constant g0 = N_k (fastmax_hack.py:17-31)."""

    @staticmethod
    def forward(ctx, q, k, v, g0):
        q, k, v = q.float(), k.float(), v.float()
        o, g = ops.forward(ops._prep(q, q.device), ops._prep(k, q.device), ops._prep(v, q.device), 1, False, 1.0,
                           g0, torch.float32)
        ctx.save_for_backward(q, k, v, o, g)
        return o

    @staticmethod
    def backward(ctx, go):
        q, k, v, o, g = ctx.saved_tensors
        dq, dk, dv = ops.backward(ops._prep(q, q.device), ops._prep(k, q.device), ops._prep(v, q.device), o, g,
                                  ops._prep(go.float(), q.device), 1, False, 1.0)
        return dq, dk, dv, None


def _unmasked_first_order(qn, kn, v, g0):
    return _UnmaskedNk.apply(qn, kn, v, g0)
