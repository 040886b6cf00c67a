"""Host side of the fastmax operator: tensor plumbing around the C ABI (include/fastmax_hip.h).

PyTorch is used for device memory, streams and autograd bookkeeping only; all arithmetic of
the hot path runs in libfastmax_hip.so.
"""
import ctypes
import os
import math

import torch

from . import _lib

_DT = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16, torch.float16: _lib.F16}
_force_path = _lib.PATH_AUTO


def set_forced_path(path):
    """Testing / benchmarking hook: force a kernel family (``_lib.PATH_*``); AUTO restores dispatch."""
    global _force_path
    _force_path = int(path)


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("fastmax needs an MI355X (HIP device); there is no CPU fallback for this operator")
    return torch.device("cuda", torch.cuda.current_device())


def _prep(t, dev):
    """-> tensor on the HIP device with unit stride in D and 16-byte aligned rows."""
    if t.device.type != "cuda":
        t = t.to(dev, non_blocking=False)
    es = t.element_size()
    ok = t.stride(3) == 1 and all((s * es) % 16 == 0 for s in t.stride()[:3]) and t.data_ptr() % 16 == 0
    return t if ok else t.contiguous()


def _strides(t):
    return (ctypes.c_int64 * 3)(t.stride(0), t.stride(1), t.stride(2))


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _ws(nbytes, dev):
    if nbytes == 0:
        return None, ctypes.c_void_p(0)
    buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    return buf, ctypes.c_void_p(buf.data_ptr())


def _problem(q, k, in_dt, out_dt, p, causal, nt, g0):
    B, H, Nq, D = q.shape
    return _lib.Problem(B, H, Nq, k.shape[2], D, _DT[in_dt], _DT[out_dt], int(p), int(causal), 1.0 / nt,
                        1.0 / (2.0 * nt * nt), float(g0), _force_path)


def selected_path(q, k, p, causal, nt=1.0):
    in_dt = q.dtype if q.dtype in _DT else torch.float32
    out_dt = in_dt if causal else torch.float32          # dtype rule Q1
    prob = _problem(q, k, in_dt, out_dt, p, causal, nt, 0.0)
    return _lib.lib().fastmax_hip_select_path(ctypes.byref(prob))


KEEP_STATES = os.environ.get("FASTMAX_KEEP_STATES", "1") != "0"


MAX_HEAD_SIZE = 256      # FASTMAX_MAX_D (include/fastmax_hip.h): the largest head size in lit_gpt/config.py (pythia-1b, Gemma-2b)


def forward(q, k, v, p, causal, nt, g0, out_dtype, need_g=True, keep_states=False):
    """q,k,v: device tensors (B,H,N,D) of one dtype in {f32,bf16,f16}. -> (o, g), or (o, g, states) with ``keep_states``:
    the sequence-split prefix states the p=1 masked forward left in its workspace (None when this call has none), for
    ``backward(..., states=...)``"""
    L = _lib.lib()
    if p not in (1, 2):
        raise ValueError(f"p should be 1 or 2, got p={p}")
    dev = q.device
    B, H, Nq, D = q.shape
    if D > MAX_HEAD_SIZE:
        raise NotImplementedError(f"head size {D} > {MAX_HEAD_SIZE} is not supported by the HIP kernels")
    prob = _problem(q, k, q.dtype, out_dtype, p, causal, nt, g0)
    o = torch.empty((B, H, Nq, D), dtype=out_dtype, device=dev)
    g = torch.empty((B, H, Nq), dtype=torch.float32, device=dev) if need_g else None
    wsb, wsp = _ws(L.fastmax_hip_forward_workspace(ctypes.byref(prob)), dev)
    with torch.cuda.device(dev):
        rc = L.fastmax_hip_forward(ctypes.byref(prob), q.data_ptr(), _strides(q), k.data_ptr(), _strides(k),
                                   v.data_ptr(), _strides(v), o.data_ptr(), g.data_ptr() if need_g else None,
                                   wsp, wsb.numel() if wsb is not None else 0, _stream(dev))
    _lib.check(rc, "fastmax_hip_forward")
    if keep_states:
        nb = 0
        if KEEP_STATES and wsb is not None:
            nb = L.fastmax_hip_forward_state_bytes(ctypes.byref(prob), q.data_ptr(), _strides(q), k.data_ptr(), _strides(k),
                                                   v.data_ptr(), _strides(v), o.data_ptr())
        return o, g, (wsb if 0 < nb <= wsb.numel() else None)
    return o, g


def backward(q, k, v, o, g, grad_o, p, causal, nt, states=None):
    L = _lib.lib()
    dev = q.device
    prob = _problem(q, k, q.dtype, o.dtype, p, causal, nt, 0.0)
    dq = torch.empty(q.shape, dtype=q.dtype, device=dev)
    dk = torch.empty(k.shape, dtype=q.dtype, device=dev)
    dv = torch.empty(v.shape, dtype=q.dtype, device=dev)
    wsb, wsp = _ws(L.fastmax_hip_backward_workspace(ctypes.byref(prob)), dev)
    with torch.cuda.device(dev):
        rc = L.fastmax_hip_backward_with_states(ctypes.byref(prob), q.data_ptr(), _strides(q), k.data_ptr(), _strides(k),
                                                v.data_ptr(), _strides(v), o.data_ptr(), g.data_ptr(), grad_o.data_ptr(),
                                                _strides(grad_o), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), wsp,
                                                wsb.numel() if wsb is not None else 0,
                                                None if states is None else states.data_ptr(),
                                                0 if states is None else states.numel(), _stream(dev))
    _lib.check(rc, "fastmax_hip_backward_with_states")
    return dq, dk, dv


def normalize(x):
    """linearmax prologue on the device. x: (B,H,N,D) -> (y float32 contiguous, inv_norm (B,H) float32)."""
    L = _lib.lib()
    dev = x.device
    B, H, N, D = x.shape
    if D > MAX_HEAD_SIZE:
        raise NotImplementedError(f"head size {D} > {MAX_HEAD_SIZE} is not supported by the HIP kernels")
    y = torch.empty((B, H, N, D), dtype=torch.float32, device=dev)
    inv = torch.empty((B, H), dtype=torch.float32, device=dev)
    wsb, wsp = _ws(L.fastmax_hip_normalize_workspace(B, H), dev)
    with torch.cuda.device(dev):
        rc = L.fastmax_hip_normalize(x.data_ptr(), _strides(x), _DT[x.dtype], y.data_ptr(), inv.data_ptr(), B, H, N,
                                     D, wsp, wsb.numel(), _stream(dev))
    _lib.check(rc, "fastmax_hip_normalize")
    return y, inv


def normalize_cast(x, rep=1):
    """linearmax prologue written in x's own dtype (training route): -> (y like x, contiguous; inv_norm (B,H) float32),
    or None when the head size is not a whole number of 16-byte pieces (caller falls back to normalize()).
    rep > 1 (grouped-query attention): x holds the G key heads, y the G * rep query-head copies (B, G * rep, N, D)."""
    L = _lib.lib()
    dev = x.device
    B, H, N, D = x.shape
    y = torch.empty((B, H * rep, N, D), dtype=x.dtype, device=dev)
    inv = torch.empty((B, H), dtype=torch.float32, device=dev)
    # room for one word per 256-token block of every head: the two-launch form (see include/fastmax_hip.h)
    wsb, wsp = _ws(max(L.fastmax_hip_normalize_workspace(B, H), 4 * B * H * ((N + 255) // 256)), dev)
    with torch.cuda.device(dev):
        if rep == 1:
            rc = L.fastmax_hip_normalize_cast(x.data_ptr(), _strides(x), _DT[x.dtype], y.data_ptr(), inv.data_ptr(), B, H, N, D,
                                              wsp, wsb.numel(), _stream(dev))
        else:
            rc = L.fastmax_hip_normalize_cast_expand(x.data_ptr(), _strides(x), _DT[x.dtype], y.data_ptr(), inv.data_ptr(), B, H,
                                                     rep, N, D, wsp, wsb.numel(), _stream(dev))
    if rc == _lib.E_BAD_SHAPE:
        return None
    _lib.check(rc, "fastmax_hip_normalize_cast")
    return y, inv


def normalize_backward(x, gy, inv, rep=1):
    """gradient of normalize_cast wrt x; gy like y (made contiguous), inv from the forward."""
    L = _lib.lib()
    dev = x.device
    B, H, N, D = x.shape
    gy = gy.to(x.dtype).contiguous()
    gx = torch.empty((B, H, N, D), dtype=x.dtype, device=dev)
    wsb, wsp = _ws(L.fastmax_hip_normalize_backward_workspace(B, H * rep, N), dev)
    with torch.cuda.device(dev):
        if rep == 1:
            rc = L.fastmax_hip_normalize_backward(x.data_ptr(), _strides(x), _DT[x.dtype], gy.data_ptr(), inv.data_ptr(),
                                                  gx.data_ptr(), B, H, N, D, wsp, wsb.numel(), _stream(dev))
        else:
            rc = L.fastmax_hip_normalize_backward_expand(x.data_ptr(), _strides(x), _DT[x.dtype], gy.data_ptr(), inv.data_ptr(),
                                                         gx.data_ptr(), B, H, rep, N, D, wsp, wsb.numel(), _stream(dev))
    _lib.check(rc, "fastmax_hip_normalize_backward")
    return gx


def normalize_stats(x):
    """1 / max_n ||x_n - mean_D x_n|| per (b,h) -- the only global quantity of the linearmax prologue."""
    L = _lib.lib()
    dev = x.device
    B, H, N, D = x.shape
    inv = torch.empty((B, H), dtype=torch.float32, device=dev)
    wsb, wsp = _ws(L.fastmax_hip_normalize_workspace(B, H), dev)
    with torch.cuda.device(dev):
        rc = L.fastmax_hip_normalize_stats(x.data_ptr(), _strides(x), _DT[x.dtype], inv.data_ptr(), B, H, N, D, wsp,
                                           wsb.numel(), _stream(dev))
    _lib.check(rc, "fastmax_hip_normalize_stats")
    return inv


def normalize_stats_pair(q, k):
    """normalize_stats of q and of k in two launches in all (per-block maxima of both tensors, one fold)"""
    if q.shape != k.shape or q.dtype != k.dtype or q.device != k.device or q.shape[0] * q.shape[1] > 65535:
        return normalize_stats(q), normalize_stats(k)
    L = _lib.lib()
    dev = q.device
    B, H, N, D = q.shape
    qi = torch.empty((B, H), dtype=torch.float32, device=dev)
    ki = torch.empty((B, H), dtype=torch.float32, device=dev)
    wsb, wsp = _ws(L.fastmax_hip_normalize_stats2_workspace(B, H, N), dev)
    with torch.cuda.device(dev):
        rc = L.fastmax_hip_normalize_stats2(q.data_ptr(), _strides(q), k.data_ptr(), _strides(k), _DT[q.dtype], qi.data_ptr(),
                                            ki.data_ptr(), B, H, N, D, wsp, wsb.numel(), _stream(dev))
    _lib.check(rc, "fastmax_hip_normalize_stats2")
    return qi, ki


def linearmax_forward_fused(q, k, v, return_stats=False, train=False):
    """Masked first-order linearmax with the prologue fused into the matrix-core kernel.
    Returns None when the shape / dtype is not covered (the caller then uses the unfused route).
    ``train``: -> (o, g, inv_q, inv_k, states, k_nstar) for linearmax_backward (states = the forward's workspace when it holds
    the sequence split's prefix states, else None; nstar (2, B*H) = per head the row of q / of k that attains the max-norm)."""
    L = _lib.lib()
    dev = q.device
    B, H, N, D = q.shape
    if q.dtype not in _DT or D > 128:
        return None
    prob = _problem(q, k, q.dtype, q.dtype, 1, True, 1.0, 0.0)
    if B * H > 65535:
        return None
    if train and not L.fastmax_hip_linearmax_train_supported(ctypes.byref(prob)):
        return None
    # statistics + scan in one entry point (with the sequence split the statistics ride on the split's state pass)
    stats = torch.empty((2, B * H), dtype=torch.float32, device=dev)
    nstar = torch.full((2, B * H), -1, dtype=torch.int32, device=dev) if train else None
    o = torch.empty((B, H, N, D), dtype=q.dtype, device=dev)
    g = torch.empty((B, H, N), dtype=torch.float32, device=dev) if train else None
    wsb, wsp = _ws(L.fastmax_hip_linearmax_forward_auto_workspace(ctypes.byref(prob)), dev)
    with torch.cuda.device(dev):
        rc = L.fastmax_hip_linearmax_forward_auto(ctypes.byref(prob), q.data_ptr(), _strides(q), k.data_ptr(), _strides(k),
                                                  v.data_ptr(), _strides(v), stats[0].data_ptr(), stats[1].data_ptr(),
                                                  nstar[0].data_ptr() if train else None, nstar[1].data_ptr() if train else None,
                                                  o.data_ptr(), g.data_ptr() if train else None, wsp,
                                                  wsb.numel() if wsb is not None else 0, _stream(dev))
    if rc in (-2, -5):          # FASTMAX_E_BAD_SHAPE / _ALIGNMENT: not covered by the fused kernel
        return None
    _lib.check(rc, "fastmax_hip_linearmax_forward_auto")
    if train:
        nb = 0
        if KEEP_STATES and wsb is not None:
            nb = L.fastmax_hip_forward_state_bytes(ctypes.byref(prob), q.data_ptr(), _strides(q), k.data_ptr(), _strides(k),
                                                   v.data_ptr(), _strides(v), o.data_ptr())
        return o, g, stats[0], stats[1], (wsb if 0 < nb <= wsb.numel() else None), nstar
    return (o, stats[0], stats[1]) if return_stats else o


def linearmax_backward(q, k, v, o, g, grad_o, inv_q, inv_k, states=None, nstar=None, fuse=0):
    """backward of linearmax_forward_fused(train=True): q, k RAW, the scans apply the prologue while staging.
    -> (dq_n, dk_n, dv): gradients wrt the NORMALISED q, k (finish with normalize_backward) and wrt v.  With ``nstar`` (from
    the forward) and ``fuse`` bit 0 the dK/dV kernel applies the prologue's backward itself and dk_n is the gradient wrt the raw
    k; bit 1 (with bit 0): the dQ kernel does the same for dq_n."""
    L = _lib.lib()
    dev = q.device
    prob = _problem(q, k, q.dtype, o.dtype, 1, True, 1.0, 0.0)
    dq = torch.empty(q.shape, dtype=q.dtype, device=dev)
    dk = torch.empty(q.shape, dtype=q.dtype, device=dev)
    dv = torch.empty(v.shape, dtype=q.dtype, device=dev)
    wsb, wsp = _ws(L.fastmax_hip_backward_workspace(ctypes.byref(prob)), dev)
    with torch.cuda.device(dev):
        rc = L.fastmax_hip_linearmax_backward(ctypes.byref(prob), q.data_ptr(), _strides(q), k.data_ptr(), _strides(k),
                                              v.data_ptr(), _strides(v), o.data_ptr(), g.data_ptr(), grad_o.data_ptr(),
                                              _strides(grad_o), inv_q.data_ptr(), inv_k.data_ptr(),
                                              None if nstar is None else nstar[0].data_ptr(),
                                              None if nstar is None else nstar[1].data_ptr(), dq.data_ptr(), dk.data_ptr(),
                                              dv.data_ptr(), wsp, wsb.numel() if wsb is not None else 0,
                                              None if states is None else states.data_ptr(),
                                              0 if states is None else states.numel(), 0 if nstar is None else fuse, _stream(dev))
    _lib.check(rc, "fastmax_hip_linearmax_backward")
    return dq, dk, dv


def effective_normalize_term(D, normalize_term, tensors_normalized):
    # attention_mechanisms/fastmax.py:78-82
    return 1.0 if tensors_normalized is True else normalize_term * math.sqrt(D)


def rope_qkv_supported(dtype, head_size, rope_n_elem):
    e = 4 if dtype == torch.float32 else 8
    return dtype in _DT and rope_n_elem % 2 == 0 and (rope_n_elem // 2) % e == 0 and (head_size - rope_n_elem) % e == 0


_ROPE_F32 = {}


def _rope_tables_f32(cos, sin, T, n_elem):
    """the kernel's exact float32 copies of the first T rows of the rope cache; every layer of a step asks for the same
    tables, so the last conversion is kept (keyed by storage, version and shape of the cache tensors)"""
    if cos.is_cuda and torch.cuda.is_current_stream_capturing():
        # a recorded HIP graph keeps reading these tables on every replay: allocate them inside the capture, so that they
        # live in the graph's own memory pool (an entry of the one-slot cache below is freed by the next eager call with
        # another sequence length, rope cache or model)
        return cos[:T, :n_elem].float().contiguous(), sin[:T, :n_elem].float().contiguous()
    key = (cos.data_ptr(), sin.data_ptr(), cos._version, sin._version, tuple(cos.shape), cos.dtype, str(cos.device), T, n_elem)
    hit = _ROPE_F32.get("last")
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    c = cos[:T, :n_elem].float().contiguous()
    s_ = sin[:T, :n_elem].float().contiguous()
    _ROPE_F32["last"] = (key, c, s_, cos, sin)          # the cache tensors are held so that their addresses cannot be reused
    return c, s_


def rope_qkv_backward(gq, gk, gv, cos32, sin32, B, T, G, qpk, hs, rope_n, kern_expand):
    """gradient of the QKV split + RoPE pass: (B,T,G,qpk+2,hs) from the gradients of q, k, v (inverse rotation, re-interleave;
    kern_expand 1 / 2: k, v (or v) arrive per query head and are summed over their group while they are read)"""
    gq, gk, gv = gq.contiguous(), gk.contiguous(), gv.contiguous()
    gqkv = torch.empty((B, T, G, qpk + 2, hs), dtype=gq.dtype, device=gq.device)
    with torch.cuda.device(gq.device):
        rc = _lib.lib().fastmax_hip_rope_qkv_split_backward(gq.data_ptr(), gk.data_ptr(), gv.data_ptr(), cos32.data_ptr(),
                                                            sin32.data_ptr(), gqkv.data_ptr(), B, T, G, qpk, hs, rope_n, kern_expand,
                                                            _DT[gq.dtype], _stream(gq.device))
    _lib.check(rc, "fastmax_hip_rope_qkv_split_backward")
    return gqkv


class RopeQKVSplit(torch.autograd.Function):
    """qkv (B,T,G,q_per_kv+2,hs) -> q (B,H,T,hs), k, v (B,H,T,hs): de-interleave + RoPE + GQA expand in one HIP pass
    (fastmax_rope.hip; lit_gpt/model.py:397-425), and the mirror pass for the gradient."""

    @staticmethod
    def forward(ctx, qkv, cos, sin, rope_n_elem, expand=1):
        """expand: what the key / value heads look like to the attention that follows (H = G * q_per_kv query heads):
          0  k, v stay at their G heads (B,G,T,hs)
          1  k, v COPIED for the query heads of their group (B,H,T,hs) -- the reference's expand + reshape, model.py:404-420
          2  k stays (B,G,T,hs), v copied
          3  nothing is copied: q comes back as (B*G, q_per_kv, T, hs) and k, v as views of the G-head tensors with head
             stride 0, shaped the same -- (batch, group) becomes the batch axis and the kernels' own strides do the
             group indexing, so every query head of a group reads the same K, V rows; the gradients then arrive per
             query head and the backward pass sums them over the group while it reads them
          4  k stays (B,G,T,hs) (for the linearmax prologue, which normalises per key head), q and v as in 3"""
        L = _lib.lib()
        B, T, G, total, hs = qkv.shape
        qpk = total - 2
        qkv = qkv.contiguous()
        # a rope cache kept in the tensors' own 16-bit dtype ("bf16-true"): model.py:708 then rounds each product to that
        # dtype before the sum -- the kernel reproduces those roundings (tables travel as exact float32 copies)
        tables16 = 16 if (cos.dtype == qkv.dtype and qkv.dtype in (torch.bfloat16, torch.float16)) else 0
        cos, sin = _rope_tables_f32(cos, sin, T, rope_n_elem)
        k_copies, v_copies = expand == 1, expand in (1, 2)
        kern_expand = 1 if (k_copies and v_copies) else (2 if v_copies else 0)          # what the forward pass materialises
        q = torch.empty((B, G * qpk, T, hs), dtype=qkv.dtype, device=qkv.device)
        k = torch.empty((B, G * qpk if k_copies else G, T, hs), dtype=qkv.dtype, device=qkv.device)
        v = torch.empty((B, G * qpk if v_copies else G, T, hs), dtype=qkv.dtype, device=qkv.device)
        with torch.cuda.device(qkv.device):
            rc = L.fastmax_hip_rope_qkv_split(qkv.data_ptr(), cos.data_ptr(), sin.data_ptr(), q.data_ptr(), k.data_ptr(),
                                              v.data_ptr(), B, T, G, qpk, hs, rope_n_elem, kern_expand | tables16, _DT[qkv.dtype],
                                              _stream(qkv.device))
        _lib.check(rc, "fastmax_hip_rope_qkv_split")
        ctx.save_for_backward(cos, sin)
        ctx.dims = (B, T, G, qpk, hs, rope_n_elem, int(expand))
        if expand in (3, 4):
            def group_view(t):
                return t.view(B * G, 1, T, hs).expand(B * G, qpk, T, hs)
            return q.view(B * G, qpk, T, hs), (group_view(k) if expand == 3 else k), group_view(v)
        return q, k, v

    @staticmethod
    def backward(ctx, gq, gk, gv):
        L = _lib.lib()
        cos, sin = ctx.saved_tensors
        B, T, G, qpk, hs, rope_n_elem, expand = ctx.dims
        # modes 3 / 4: gradients of the stride-0 views arrive dense, one per query head, laid out (B*G, qpk, T, hs) =
        # (B, H, T, hs): exactly what the kernel's group-summing read expects for copied heads
        kern_expand = {0: 0, 1: 1, 2: 2, 3: 1, 4: 2}[expand]
        gq, gk, gv = gq.contiguous(), gk.contiguous(), gv.contiguous()
        gqkv = torch.empty((B, T, G, qpk + 2, hs), dtype=gq.dtype, device=gq.device)
        with torch.cuda.device(gq.device):
            rc = L.fastmax_hip_rope_qkv_split_backward(gq.data_ptr(), gk.data_ptr(), gv.data_ptr(), cos.data_ptr(), sin.data_ptr(),
                                                       gqkv.data_ptr(), B, T, G, qpk, hs, rope_n_elem, kern_expand, _DT[gq.dtype],
                                                       _stream(gq.device))
        _lib.check(rc, "fastmax_hip_rope_qkv_split_backward")
        return gqkv, None, None, None, None
