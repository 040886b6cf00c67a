"""Opt-in decode-time state cache for first-order fastmax.

The reference generates with a zero-padded KV cache and re-runs UNMASKED attention over the whole cache for every
new token (lit_gpt/model.py:427-430, 464-466; generate/base.py:85-92): O(N D) per token and, by quirk Q4, not the
same function as masked attention over the real sequence.  Here the carried state (S2 = sum k v^T, S1 = sum v,
ksum = sum k, count) lives in HBM: `prefill` builds it from the prompt, `step` costs O(D^2) per token and returns
masked p=1 fastmax at the new last position.  Because it changes the decode semantics it is a separate class, not
a silent replacement of `fastmax(mask=False)`.
"""
import ctypes
import math

import torch

from . import _lib, ops
from .attention_mechanisms.fastmax import fastmax


class FastmaxDecodeState:
    def __init__(self, B, H, D, device, normalize_term=8, tensors_normalized=False):
        self.B, self.H, self.D = B, H, D
        self.nt = ops.effective_normalize_term(D, normalize_term, tensors_normalized)
        self._kw = dict(normalize_term=normalize_term, tensors_normalized=tensors_normalized)
        nbytes = _lib.lib().fastmax_hip_decode_state_bytes(B, H, D)
        if nbytes == 0:
            raise NotImplementedError(f"head size {D} not supported")
        self.state = torch.zeros(nbytes // 4, dtype=torch.float32, device=device)
        self.count = 0

    def prefill(self, q, k, v):
        """Masked p=1 forward over the prompt; also captures the end-of-prompt state.  Returns o (B,H,N,D)."""
        assert self.count == 0, "prefill starts a sequence"
        L = _lib.lib()
        o = fastmax(q, k, v, mask=True, p=1, **self._kw)
        kd, vd = ops._prep(k, k.device), ops._prep(v, v.device)
        prob = ops._problem(kd, kd, kd.dtype, kd.dtype, 1, True, self.nt, 0.0)
        with torch.cuda.device(kd.device):
            rc = L.fastmax_hip_p1_prefill_state(ctypes.byref(prob), kd.data_ptr(), ops._strides(kd), vd.data_ptr(),
                                                ops._strides(vd), self.state.data_ptr(), ops._stream(kd.device))
        _lib.check(rc, "fastmax_hip_p1_prefill_state")
        self.count = k.shape[2]
        return o

    def step(self, q, k, v):
        """q,k,v: (B,H,1,D) of the new token -> o (B,H,1,D); O(D^2) per head."""
        L = _lib.lib()
        qd, kd, vd = (ops._prep(t, t.device) for t in (q, k, v))
        self.count += 1
        o = torch.empty((self.B, self.H, 1, self.D), dtype=q.dtype, device=q.device)
        dt = ops._DT[q.dtype]
        with torch.cuda.device(q.device):
            rc = L.fastmax_hip_p1_decode_step(qd.data_ptr(), ops._strides(qd), kd.data_ptr(), ops._strides(kd), vd.data_ptr(),
                                              ops._strides(vd), self.state.data_ptr(), o.data_ptr(), self.B, self.H, self.D,
                                              dt, dt, 1.0 / self.nt, self.count, ops._stream(q.device))
        _lib.check(rc, "fastmax_hip_p1_decode_step")
        return o
