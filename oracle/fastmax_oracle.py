"""CPU oracle for the fastmax / linearmax polynomial-attention path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``fastmax_experiments_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` do, and there only as the checker / the timed CPU baseline.

What it restates (reference = /root/reference, read-only, never shipped):

* ``attention_mechanisms/fastmax.py:41-111``  forward (``F/g``), ``normalize_term`` rule 78-82
* ``attention_mechanisms/fastmax.py:184-322``  ``compute_F_*`` / ``compute_g_*`` (masked + unmasked)
* ``attention_mechanisms/fastmax.py:383-691``  the six ``gradient_o_{q,k,v}_{masked,unmasked}``
* ``attention_mechanisms/fastmax.py:326-334``  ``normalize``
* ``attention_mechanisms/fastmax.py:336-381``  ``compute_attn`` (dense known-answer path)
* ``attention_mechanisms/fastmax_hack.py:5-60`` linearmax, masked and unmasked branches

Parity pinning: the reference holds no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference itself,
generated in the build container by ``tests/golden/make_golden.py`` and committed as
``tests/golden/*.npz`` (``tests/test_oracle_golden.py`` checks every one of them).

The restatement is NOT a transcription of the einops code.  It uses the unified form
(SURVEY.md section 8a):   with V' = [V | 1]  (ones column appended),

    [F | g]_i = S1 + a * q_i^T S2 + b * (q_i (x) q_i) : S3
    S1 = sum v'_j ,  S2 = sum k_j (x) v'_j ,  S3 = sum k_j (x) k_j (x) v'_j

with the sums running over j <= i (masked) or all j (unmasked), a = 1/nt,
b = 1/(2 nt^2) (p = 2 only), evaluated chunk by chunk with the state carried
(``*_factorized``), plus a dense O(N^2) evaluation (``*_dense``) of the same function.
All arithmetic is numpy in the dtype asked for (float64 by default).
"""
from __future__ import annotations

import math

import numpy as np

__all__ = [
    "effective_normalize_term",
    "fastmax_fwd_factorized",
    "fastmax_fwd_dense",
    "fastmax_bwd_factorized",
    "fastmax_bwd_dense",
    "normalize_qk",
    "linearmax_fwd",
    "compute_attn_dense",
    "fastmax_fwd_reference_equivalent",
]


def effective_normalize_term(D: int, normalize_term=8, tensors_normalized=False) -> float:
    """fastmax.py:78-82 -- nt = 1 if tensors_normalized else normalize_term*sqrt(D)."""
    if tensors_normalized is True:
        return 1.0
    return float(normalize_term) * math.sqrt(D)


def _check_p(p):
    # fastmax.py:362,428,483,... raise ValueError for p outside {1,2}
    if p not in (1, 2):
        raise ValueError(f"p should be 1 or 2, got p={p}")


def _f(s, p):
    return 1.0 + s if p == 1 else 1.0 + s + 0.5 * s * s


def _p2_chunk(chunk, D, p):
    # keep the per-chunk (chunk, D, D, D+1) temporaries of the p=2 path around 32 MiB
    return chunk if p == 1 else max(1, min(chunk, (1 << 22) // (D ** 3)))


def _fprime(s, p):
    return np.ones_like(s) if p == 1 else 1.0 + s


# --------------------------------------------------------------------------------------
# forward
# --------------------------------------------------------------------------------------
def fastmax_fwd_dense(q, k, v, mask=True, nt=None, p=1, g_const=None, dtype=np.float64):
    """Dense evaluation of the operator: o_i = sum_j f(s_ij) v_j / sum_j f(s_ij).

    Known-answer form of fastmax.py:336-381 (``compute_attn``) followed by 103.
    ``g_const``: constant term of the denominator in the UNMASKED case.  The reference
    uses N_q there (fastmax.py:269-271) while fastmax_hack.py:21 uses N_k; they only
    differ when N_q != N_k.  None -> N_q (fastmax.py behaviour).
    Returns (o, g) with g the denominator, shape (B,H,Nq).
    """
    _check_p(p)
    q = np.asarray(q, dtype=dtype)
    k = np.asarray(k, dtype=dtype)
    v = np.asarray(v, dtype=dtype)
    D = q.shape[-1]
    nt = effective_normalize_term(D) if nt is None else nt
    Nq, Nk = q.shape[2], k.shape[2]
    s = np.einsum("bhid,bhjd->bhij", q, k) / nt
    P = _f(s, p)
    if mask:
        assert Nq == Nk
        P = P * np.tril(np.ones((Nq, Nk), dtype=dtype))
        g = P.sum(-1)
    else:
        # rowsum(P) has constant term N_k; the reference's is g_const (default N_q)
        gc = Nq if g_const is None else g_const
        g = P.sum(-1) - Nk + gc
    F = np.einsum("bhij,bhjd->bhid", P, v)
    return F / g[..., None], g


def fastmax_fwd_factorized(q, k, v, mask=True, nt=None, p=1, g_const=None, chunk=32,
                           dtype=np.float64):
    """Factorised (linear in N) evaluation with carried state, chunk by chunk.

    Follows fastmax.py:218-250 (compute_F_masked), 287-322 (compute_g_masked),
    184-216 / 252-285 (unmasked), 97 (o = F/g).  Returns (o, g).
    """
    _check_p(p)
    q = np.asarray(q, dtype=dtype)
    k = np.asarray(k, dtype=dtype)
    v = np.asarray(v, dtype=dtype)
    B, H, Nq, D = q.shape
    Nk = k.shape[2]
    nt = effective_normalize_term(D) if nt is None else nt
    a = 1.0 / nt
    b = 1.0 / (2.0 * nt * nt)
    vp = np.concatenate([v, np.ones((B, H, Nk, 1), dtype=dtype)], axis=-1)  # V' = [V | 1]
    out = np.empty((B, H, Nq, D + 1), dtype=dtype)
    chunk = _p2_chunk(chunk, D, p)

    if not mask:
        S1 = vp.sum(2)                                           # (B,H,D+1)
        S2 = np.einsum("bhjm,bhjr->bhmr", k, vp)                 # (B,H,D,D+1)
        out[:] = S1[:, :, None, :] + a * np.einsum("bhim,bhmr->bhir", q, S2)
        if p == 2:
            S3 = np.einsum("bhjm,bhjl,bhjr->bhmlr", k, k, vp)    # (B,H,D,D,D+1)
            out += b * np.einsum("bhim,bhil,bhmlr->bhir", q, q, S3)
        # fastmax.py:269-271: constant term of g is N_q (not the number of keys)
        gc = Nq if g_const is None else g_const
        out[..., D] += gc - Nk
    else:
        assert Nq == Nk
        S1 = np.zeros((B, H, D + 1), dtype=dtype)
        S2 = np.zeros((B, H, D, D + 1), dtype=dtype)
        S3 = np.zeros((B, H, D, D, D + 1), dtype=dtype) if p == 2 else None
        for c0 in range(0, Nq, chunk):
            c1 = min(Nq, c0 + chunk)
            qc, kc, vc = q[:, :, c0:c1], k[:, :, c0:c1], vp[:, :, c0:c1]
            # contribution of all earlier chunks through the carried state
            acc = S1[:, :, None, :] + a * np.einsum("bhim,bhmr->bhir", qc, S2)
            if p == 2:
                acc = acc + b * np.einsum("bhim,bhil,bhmlr->bhir", qc, qc, S3)
            # inside the chunk: causal prefix sums of the per-token outer products
            acc = acc + np.cumsum(vc, axis=2)
            kv = np.einsum("bhjm,bhjr->bhjmr", kc, vc)
            acc = acc + a * np.einsum("bhim,bhimr->bhir", qc, np.cumsum(kv, axis=2))
            if p == 2:
                kkv = np.einsum("bhjm,bhjl,bhjr->bhjmlr", kc, kc, vc)
                acc = acc + b * np.einsum("bhim,bhil,bhimlr->bhir", qc, qc,
                                          np.cumsum(kkv, axis=2))
                S3 = S3 + kkv.sum(2)
            out[:, :, c0:c1] = acc
            S1 = S1 + vc.sum(2)
            S2 = S2 + kv.sum(2)
    g = out[..., D].copy()
    return out[..., :D] / g[..., None], g


def fastmax_fwd_reference_equivalent(q, k, v, nt, p=1):
    """Masked forward with the reference's OP STRUCTURE (materialise the per-token outer
    products over the whole sequence, then cumsum): fastmax.py:236-248, 306-320.  Used
    only as the 'what the reference CPU path costs' timing baseline (torch CPU ops, so
    it threads like the reference does); numerically the same function as above.
    """
    import torch

    _check_p(p)
    a = 1.0 / nt
    F = torch.cumsum(v, 2)
    kv = torch.einsum("bhnm,bhnj->bhnmj", k, v)
    F = F + a * torch.einsum("bhim,bhimj->bhij", q, torch.cumsum(kv, 2))
    N = q.shape[2]
    g = (torch.arange(N, device=q.device) + 1)[None, None, :] \
        + a * torch.einsum("bhim,bhim->bhi", q, torch.cumsum(k, 2))
    if p == 2:
        b = 1.0 / (2 * nt * nt)
        kkv = torch.einsum("bhnm,bhnl,bhnj->bhnmlj", k, k, v)
        F = F + b * torch.einsum("bhim,bhil,bhimlj->bhij", q, q, torch.cumsum(kkv, 2))
        kk = torch.einsum("bhnm,bhnl->bhnml", k, k)
        g = g + b * torch.einsum("bhim,bhil,bhiml->bhi", q, q, torch.cumsum(kk, 2))
    return F / g[..., None]


# --------------------------------------------------------------------------------------
# backward
# --------------------------------------------------------------------------------------
def fastmax_bwd_dense(q, k, v, grad_o, mask=True, nt=None, p=1, g_const=None,
                      dtype=np.float64):
    """Gradients of L wrt q,k,v given G = dL/do, by differentiating the dense form.

    With P = f(s)*mask, g = rowsum(P) (+ const), o = P v / g, c_i = G_i . o_i:
        dP_ij = (G_i . v_j - c_i) / g_i ;  dS = dP * f'(s) * mask
        dQ = dS K / nt ; dK = dS^T Q / nt ; dV = (P/g)^T G
    This is what autograd of fastmax.py:336-381 + 103 gives and what the hand-derived
    fastmax.py:383-691 must equal.  Returns (dq, dk, dv).
    """
    _check_p(p)
    q = np.asarray(q, dtype=dtype)
    k = np.asarray(k, dtype=dtype)
    v = np.asarray(v, dtype=dtype)
    G = np.asarray(grad_o, dtype=dtype)
    D = q.shape[-1]
    nt = effective_normalize_term(D) if nt is None else nt
    Nq, Nk = q.shape[2], k.shape[2]
    s = np.einsum("bhid,bhjd->bhij", q, k) / nt
    P = _f(s, p)
    Pp = _fprime(s, p)
    if mask:
        tri = np.tril(np.ones((Nq, Nk), dtype=dtype))
        P = P * tri
        Pp = Pp * tri
        g = P.sum(-1)
    else:
        gc = Nq if g_const is None else g_const
        g = P.sum(-1) - Nk + gc
    o = np.einsum("bhij,bhjd->bhid", P, v) / g[..., None]
    c = (G * o).sum(-1)
    dP = (np.einsum("bhid,bhjd->bhij", G, v) - c[..., None]) / g[..., None]
    dS = dP * Pp
    dq = np.einsum("bhij,bhjd->bhid", dS, k) / nt
    dk = np.einsum("bhij,bhid->bhjd", dS, q) / nt
    dv = np.einsum("bhij,bhid->bhjd", P / g[..., None], G)
    return dq, dk, dv


def fastmax_bwd_factorized(q, k, v, grad_o, mask=True, nt=None, p=1, g_const=None,
                           chunk=32, dtype=np.float64):
    """Factorised backward with carried prefix (dQ) and suffix (dK, dV) states.

    Follows fastmax.py:432-485 (dQ masked: forward prefix sums), 541-604 and 647-691
    (dK, dV masked: reverse cumulative sums, there done by index reversal 565-567),
    and 383-430 / 487-539 / 606-645 (unmasked).  Unified form used here, with
    w_i = 1/g_i, c_i = G_i.o_i, Ghat_i = w_i [G_i | -c_i], v'_j = [v_j | 1]:

        dQ_i = a S2_i Ghat_i              + [p=2]  2b (q_i : S3_i) Ghat_i
        dK_j = a R2_j v'_j                + [p=2]  2b (k_j : R3_j) v'_j
        dV_j = (R1_j + a R2_j^T k_j + [p=2] b (k_j (x) k_j) : R3_j) [:D]
        S2_i = sum_{j<=i} k_j (x) v'_j,   R2_j = sum_{i>=j} q_i (x) Ghat_i   (and the
        third-order analogues), R1_j = sum_{i>=j} Ghat_i.
    """
    _check_p(p)
    q = np.asarray(q, dtype=dtype)
    k = np.asarray(k, dtype=dtype)
    v = np.asarray(v, dtype=dtype)
    G = np.asarray(grad_o, dtype=dtype)
    B, H, Nq, D = q.shape
    Nk = k.shape[2]
    nt = effective_normalize_term(D) if nt is None else nt
    a = 1.0 / nt
    b = 1.0 / (2.0 * nt * nt)
    o, g = fastmax_fwd_factorized(q, k, v, mask=mask, nt=nt, p=p, g_const=g_const,
                                  chunk=chunk, dtype=dtype)
    chunk = _p2_chunk(chunk, D, p)
    w = 1.0 / g
    c = (G * o).sum(-1)
    Gh = np.concatenate([G, -c[..., None]], axis=-1) * w[..., None]      # (B,H,Nq,D+1)
    vp = np.concatenate([v, np.ones((B, H, Nk, 1), dtype=dtype)], axis=-1)

    dq = np.empty_like(q)
    dk = np.empty_like(k)
    dv = np.empty_like(v)
    if not mask:
        S2 = np.einsum("bhjm,bhjr->bhmr", k, vp)
        R1 = Gh.sum(2)
        R2 = np.einsum("bhim,bhir->bhmr", q, Gh)
        dq[:] = a * np.einsum("bhmr,bhir->bhim", S2, Gh)
        dk[:] = a * np.einsum("bhmr,bhjr->bhjm", R2, vp)
        dvp = R1[:, :, None, :] + a * np.einsum("bhmr,bhjm->bhjr", R2, k)
        if p == 2:
            S3 = np.einsum("bhjm,bhjl,bhjr->bhmlr", k, k, vp)
            R3 = np.einsum("bhim,bhil,bhir->bhmlr", q, q, Gh)
            dq += 2 * b * np.einsum("bhil,bhmlr,bhir->bhim", q, S3, Gh)
            dk += 2 * b * np.einsum("bhjl,bhmlr,bhjr->bhjm", k, R3, vp)
            dvp = dvp + b * np.einsum("bhjm,bhjl,bhmlr->bhjr", k, k, R3)
        dv[:] = dvp[..., :D]
        return dq, dk, dv

    assert Nq == Nk
    N = Nq
    # forward scan: dQ
    S2 = np.zeros((B, H, D, D + 1), dtype=dtype)
    S3 = np.zeros((B, H, D, D, D + 1), dtype=dtype) if p == 2 else None
    for c0 in range(0, N, chunk):
        c1 = min(N, c0 + chunk)
        qc, kc, vc, gc = q[:, :, c0:c1], k[:, :, c0:c1], vp[:, :, c0:c1], Gh[:, :, c0:c1]
        kv = np.einsum("bhjm,bhjr->bhjmr", kc, vc)
        S2i = S2[:, :, None] + np.cumsum(kv, axis=2)                    # inclusive prefix
        dq[:, :, c0:c1] = a * np.einsum("bhimr,bhir->bhim", S2i, gc)
        S2 = S2 + kv.sum(2)
        if p == 2:
            kkv = np.einsum("bhjm,bhjl,bhjr->bhjmlr", kc, kc, vc)
            S3i = S3[:, :, None] + np.cumsum(kkv, axis=2)
            dq[:, :, c0:c1] += 2 * b * np.einsum("bhil,bhimlr,bhir->bhim", qc, S3i, gc)
            S3 = S3 + kkv.sum(2)
    # reverse scan: dK, dV  (suffix sums over i >= j)
    R1 = np.zeros((B, H, D + 1), dtype=dtype)
    R2 = np.zeros((B, H, D, D + 1), dtype=dtype)
    R3 = np.zeros((B, H, D, D, D + 1), dtype=dtype) if p == 2 else None
    starts = list(range(0, N, chunk))
    for c0 in reversed(starts):
        c1 = min(N, c0 + chunk)
        qc, kc, vc, gc = q[:, :, c0:c1], k[:, :, c0:c1], vp[:, :, c0:c1], Gh[:, :, c0:c1]
        rc = lambda x: np.flip(np.cumsum(np.flip(x, 2), axis=2), 2)     # inclusive suffix
        qg = np.einsum("bhim,bhir->bhimr", qc, gc)
        R1j = R1[:, :, None] + rc(gc)
        R2j = R2[:, :, None] + rc(qg)
        dk[:, :, c0:c1] = a * np.einsum("bhjmr,bhjr->bhjm", R2j, vc)
        dvp = R1j + a * np.einsum("bhjmr,bhjm->bhjr", R2j, kc)
        R1 = R1 + gc.sum(2)
        R2 = R2 + qg.sum(2)
        if p == 2:
            qqg = np.einsum("bhim,bhil,bhir->bhimlr", qc, qc, gc)
            R3j = R3[:, :, None] + rc(qqg)
            dk[:, :, c0:c1] += 2 * b * np.einsum("bhjl,bhjmlr,bhjr->bhjm", kc, R3j, vc)
            dvp = dvp + b * np.einsum("bhjm,bhjl,bhjmlr->bhjr", kc, kc, R3j)
            R3 = R3 + qqg.sum(2)
        dv[:, :, c0:c1] = dvp[..., :D]
    return dq, dk, dv


# --------------------------------------------------------------------------------------
# linearmax (fastmax_hack) and helpers
# --------------------------------------------------------------------------------------
def normalize_qk(q, k, dtype=np.float64):
    """fastmax.py:326-334 == fastmax_hack.py:38-43 (and 10-15 for the unmasked branch):
    per token subtract the mean over D; divide by the per-(b,h) max over the SEQUENCE of
    the per-token L2 norm."""
    q = np.asarray(q, dtype=dtype)
    k = np.asarray(k, dtype=dtype)
    q = q - q.mean(-1, keepdims=True)
    k = k - k.mean(-1, keepdims=True)
    qn = np.sqrt((q * q).sum(-1)).max(-1)
    kn = np.sqrt((k * k).sum(-1)).max(-1)
    return q / qn[..., None, None], k / kn[..., None, None]


def linearmax_fwd(q, k, v, p=1, mask=True, dtype=np.float64, chunk=32):
    """fastmax_hack.py:5-60.  Masked: normalise then p-th order fastmax with nt=1
    (45-56).  Unmasked (6-33): normalise, p is ignored (first order only), nt=1, and
    the denominator's constant term is N_k (line 21), not N_q."""
    qn, kn = normalize_qk(q, k, dtype=dtype)
    if mask:
        _check_p(p)
        o, _ = fastmax_fwd_factorized(qn, kn, v, mask=True, nt=1.0, p=p, chunk=chunk,
                                      dtype=dtype)
        return o
    Nk = np.asarray(k).shape[2]
    o, _ = fastmax_fwd_factorized(qn, kn, v, mask=False, nt=1.0, p=1, g_const=Nk,
                                  dtype=dtype)
    return o


def compute_attn_dense(q, k, mask=True, nt=None, p=1, dtype=np.float64):
    """fastmax.py:336-381: dense attention matrix a = f(s)*mask / rowsum."""
    _check_p(p)
    q = np.asarray(q, dtype=dtype)
    k = np.asarray(k, dtype=dtype)
    D = q.shape[-1]
    nt = effective_normalize_term(D) if nt is None else nt
    s = np.einsum("bhid,bhjd->bhij", q, k) / nt
    P = _f(s, p)
    if mask:
        P = P * np.tril(np.ones(P.shape[-2:], dtype=dtype))
    return P / P.sum(-1, keepdims=True)
