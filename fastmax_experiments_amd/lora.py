"""QLoRA linear layers for MI355X: NF4 base weight + LoRA branch, fused in one HIP kernel.

Mirrors the reference's ``lit_gpt/lora.py`` interface for the two layers on the fastmax path:
  LoRALinear       lora.py:88-177   (ctor args, lora_A/lora_B/scaling, get_lora_AB 137-139, merge 142-168, forward 170-177)
  LoRAQKVLinear    lora.py:180-433  (lora_ind 263-278, zero_pad 281-342, conv1d 344-377, get_lora_AB 379-389, forward 398-433)
and the quantised base layer the reference gets by monkey-patching ``torch.nn.Linear`` with bitsandbytes'
``Linear4bit`` under Lightning's BitsandbytesPrecision plugin (finetune/lora.py:72-78): here ``NF4Linear``
(packed uint8 ``weight`` + ``weight.quant_state``; consumers' touch points: lora.py:151-161, utils.py:36-38).

bitsandbytes is not in the reference tree and not installed: the NF4 format follows the public definition
(QLoRA, arXiv 2305.14314) -- 16-level normal-float codebook, block size 64, fp32 absmax per block, two codes
per byte with the first element in the high nibble.  Parity with bitsandbytes' own kernels is UNPINNED.

Fused forward (csrc/nf4_lora.hip):  y = x deq(W)^T + bias + (dropout(x) A^T) (scaling * scatter(B))^T
"""
import ctypes
import logging
import os
import weakref
import math
from typing import Any, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib

NF4_CODE = torch.tensor([-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453,
                         -0.28444138169288635, -0.18477343022823334, -0.09105003625154495, 0.0,
                         0.07958029955625534, 0.16093020141124725, 0.24611230194568634, 0.33791524171829224,
                         0.44070982933044434, 0.5626170039176941, 0.7229568362236023, 1.0], dtype=torch.float32)
BLOCK = 64
RANK_PAD = 32          # the fused kernel carries the LoRA branch as one 32-wide k-step


# ---------------------------------------------------------------------------------------------
# NF4 codec (host / device tensor ops; quantisation happens once at load time, not on the hot path)
# ---------------------------------------------------------------------------------------------
def nf4_quantize(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(out,in) float weight -> (packed uint8 [out*in/2], absmax float32 [out*in/64])."""
    if w.numel() % BLOCK:
        raise ValueError(f"NF4 needs a multiple of {BLOCK} weights, got {w.numel()}")
    flat = w.detach().float().reshape(-1, BLOCK)
    absmax = flat.abs().amax(dim=1)
    scaled = flat / absmax.clamp_min(1e-38)[:, None]
    code = NF4_CODE.to(w.device)
    idx = (scaled[..., None] - code).abs().argmin(dim=-1).to(torch.uint8).reshape(-1)
    packed = (idx[0::2] << 4) | idx[1::2]
    return packed.contiguous(), absmax.contiguous()


def nf4_dequantize(packed: torch.Tensor, absmax, shape, dtype=torch.float32) -> torch.Tensor:
    """Inverse of nf4_quantize (HIP kernel for device tensors, tensor ops for host tensors).  `absmax`: the fp32 block
    scales, or the weight's whole quant_state list (plain or double-quantised scales)."""
    n = int(torch.Size(shape).numel())
    qs = absmax if isinstance(absmax, (list, tuple)) else [absmax, None, None, BLOCK, None, "nf4"]
    if packed.device.type == "cuda" and dtype in (torch.float32, torch.bfloat16):
        out = torch.empty(n, dtype=dtype, device=packed.device)
        sc = NF4Scales(qs)
        with torch.cuda.device(packed.device):
            rc = _lib.lib().fastmax_hip_nf4_dequantize_s(packed.data_ptr(), sc.ref, out.data_ptr(), n,
                                                         _lib.F32 if dtype == torch.float32 else _lib.BF16,
                                                         ctypes.c_void_p(torch.cuda.current_stream(packed.device).cuda_stream))
        _lib.check(rc, "fastmax_hip_nf4_dequantize_s")
        return out.view(shape)
    code = NF4_CODE.to(packed.device)
    idx = torch.stack([packed >> 4, packed & 15], dim=1).reshape(-1).long()
    vals = code[idx].reshape(-1, BLOCK) * block_scales(qs)[:, None]
    return vals.reshape(shape).to(dtype)


# ---- double quantisation of the block scales ("bnb.nf4-dq", finetune/lora.py:38) ---------------------------------------
# QLoRA (arXiv 2305.14314, section 3): the fp32 absmax vector is shifted by its mean and stored as 8-bit codes of a
# 256-level dynamic map, blockwise with block 256 and one fp32 scale per block: 0.127 bit per weight instead of 0.5.
# bitsandbytes is absent from the reference tree and this image; the map below restates its publicly documented
# `create_dynamic_map(signed=True, max_exponent_bits=7, total_bits=8)`: for every decade 10^-6 .. 10^0 the midpoints of
# 2^i equal sub-intervals of [0.1, 1] (i = 0..6), both signs, plus 0 and 1.  Parity with bitsandbytes: UNPINNED.
DQ_BLOCK = 256


def dynamic_map_8bit() -> torch.Tensor:
    vals = [0.0, 1.0]
    for i in range(7):
        edges = torch.linspace(0.1, 1.0, 2 ** i + 1, dtype=torch.float64)
        mids = (edges[:-1] + edges[1:]) / 2 * 10.0 ** (i - 6)
        vals += mids.tolist() + (-mids).tolist()
    assert len(vals) == 256
    return torch.tensor(sorted(vals), dtype=torch.float32)


def absmax_double_quantize(absmax: torch.Tensor):
    """fp32 block scales -> (codes uint8 [n], absmax2 fp32 [ceil(n/256)], offset fp32 scalar tensor, code2 fp32 [256])."""
    a = absmax.detach().float()
    offset = a.mean()
    c = a - offset
    n = c.numel()
    pad = (-n) % DQ_BLOCK
    blocks = F.pad(c, (0, pad)).reshape(-1, DQ_BLOCK)
    absmax2 = blocks.abs().amax(dim=1)
    code2 = dynamic_map_8bit().to(a.device)
    scaled = (blocks / absmax2.clamp_min(1e-38)[:, None]).reshape(-1)[:n]
    # nearest map entry: the map is sorted, so look at the two neighbours of the insertion point
    hi = torch.searchsorted(code2, scaled).clamp(1, 255)
    lo = hi - 1
    codes = torch.where((scaled - code2[lo]).abs() <= (code2[hi] - scaled).abs(), lo, hi).to(torch.uint8)
    return codes.contiguous(), absmax2.contiguous(), offset.reshape(()), code2


def absmax_double_dequantize(codes, absmax2, offset, code2) -> torch.Tensor:
    idx = torch.arange(codes.numel(), device=codes.device) // DQ_BLOCK
    return code2[codes.long()] * absmax2[idx] + offset


class NF4Scales:
    """The block scales of one NF4 weight as the C ABI takes them (include/fastmax_hip.h `fastmax_nf4_scales`): plain fp32
    absmax, or the double-quantised form.  Keeps the tensors alive for as long as the ctypes struct is in use."""

    class _C(ctypes.Structure):
        _fields_ = [("absmax", ctypes.c_void_p), ("absmax_q", ctypes.c_void_p), ("absmax2", ctypes.c_void_p),
                    ("code2", ctypes.c_void_p), ("offset", ctypes.c_float)]

    def __init__(self, quant_state):
        stats = quant_state[4]
        if stats is None:
            self.tensors = (quant_state[0],)
            self.c = self._C(quant_state[0].data_ptr(), None, None, None, 0.0)
        else:
            offset, (absmax2, code2) = stats
            self.tensors = (quant_state[0], absmax2, code2)
            self.c = self._C(None, quant_state[0].data_ptr(), absmax2.data_ptr(), code2.data_ptr(), float(offset))
        self.ref = ctypes.byref(self.c)

    @property
    def device(self):
        return self.tensors[0].device


def scales_of(base) -> "NF4Scales":
    """the NF4Scales of an NF4Linear, rebuilt only when its quant_state tensors changed (a double-quantised state holds its
    offset as a device scalar: reading it costs a host synchronisation, which must not happen once per forward call)"""
    qs = base.weight.quant_state
    key = (qs[0].data_ptr(), None if qs[4] is None else (qs[4][1][0].data_ptr(), qs[4][1][1].data_ptr()))
    hit = base.__dict__.get("_scales_cache")
    if hit is None or hit[0] != key:
        hit = (key, NF4Scales(qs))
        base.__dict__["_scales_cache"] = hit
    return hit[1]


def block_scales(quant_state) -> torch.Tensor:
    """fp32 absmax vector of a quant_state, whichever way it is stored (host / device tensor ops)."""
    if quant_state[4] is None:
        return quant_state[0]
    offset, (absmax2, code2) = quant_state[4]
    return absmax_double_dequantize(quant_state[0], absmax2, offset.to(absmax2.device), code2)


class Params4bit(nn.Parameter):
    """Packed NF4 storage that looks like what lora.py:151-161 / utils.py:36-38 touch on a bnb weight:
    ``dtype == torch.uint8`` and ``quant_state`` with the original shape at index 1."""

    def __new__(cls, data, requires_grad=False, quant_state=None):
        # (data, requires_grad) is the call nn.Parameter.__deepcopy__ makes; a list in second place is the old
        # (data, quant_state) form
        if isinstance(requires_grad, (list, tuple)):
            requires_grad, quant_state = False, requires_grad
        self = torch.Tensor._make_subclass(cls, data, False)
        self.quant_state = quant_state
        return self

    def __deepcopy__(self, memo):
        import copy
        out = type(self)(self.data.clone(), False, copy.deepcopy(self.quant_state, memo))
        memo[id(self)] = out
        return out


class NF4Linear(nn.Module):
    """``torch.nn.Linear``-compatible frozen layer with a 4-bit NF4 weight (forward in compute dtype)."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None, dtype=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        n = in_features * out_features
        self.weight = Params4bit(torch.zeros(n // 2, dtype=torch.uint8, device=device), False,
                                 [torch.ones(n // BLOCK, dtype=torch.float32, device=device),
                                  torch.Size((out_features, in_features)), dtype or torch.bfloat16, BLOCK, None, "nf4"])
        self.bias = nn.Parameter(torch.zeros(out_features, dtype=torch.float32, device=device),
                                 requires_grad=False) if bias else None

    @classmethod
    def from_linear(cls, lin: nn.Linear, double_quant: bool = False) -> "NF4Linear":
        q = cls(lin.in_features, lin.out_features, bias=lin.bias is not None, device=lin.weight.device,
                dtype=lin.weight.dtype)
        q.load_dense(lin.weight.data, None if lin.bias is None else lin.bias.data, double_quant=double_quant)
        return q

    @property
    def double_quant(self) -> bool:
        return self.weight.quant_state[4] is not None

    def load_dense(self, w: torch.Tensor, bias=None, double_quant=None):
        """(re)quantise from a dense weight; ``double_quant`` None keeps the layer's current mode ("bnb.nf4" / "bnb.nf4-dq")"""
        if double_quant is None:
            double_quant = self.weight.quant_state is not None and self.weight.quant_state[4] is not None
        packed, absmax = nf4_quantize(w)
        qs = [absmax, torch.Size(w.shape), w.dtype, BLOCK, None, "nf4"]
        if double_quant:
            codes, absmax2, offset, code2 = absmax_double_quantize(absmax)
            qs[0], qs[4] = codes, [offset, [absmax2, code2]]
        self.weight = Params4bit(packed, False, qs)
        self._dense_cache = None
        if bias is not None:
            self.bias = nn.Parameter(bias.detach().float().clone(), requires_grad=False)

    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        self._dense_cache = None                                                          # re-enable after moving the module
        qs = self.weight.quant_state
        dev = fn(torch.empty(0, device=qs[0].device)).device                              # where `fn` sends things
        qs[0] = qs[0].to(dev)                                                             # scales keep their dtypes (fp32 / uint8)
        if qs[4] is not None:
            offset, (absmax2, code2) = qs[4]
            qs[4] = [offset.to(dev), [absmax2.to(dev), code2.to(dev)]]
        if self.weight.dtype != torch.uint8:                                              # .to(dtype) must not touch codes
            raise RuntimeError("NF4 codes were cast; move NF4Linear with .to(device) only")
        self.weight.quant_state = qs
        return self

    def __deepcopy__(self, memo):
        import copy
        out = copy.copy(self)                                   # shallow first: then every mutable member is replaced
        memo[id(self)] = out
        out._parameters = {k: copy.deepcopy(v, memo) for k, v in self._parameters.items()}
        out._buffers = {k: copy.deepcopy(v, memo) for k, v in self._buffers.items()}
        out._modules = {}
        out._dense_cache = None                                 # never share the decoded copy of the original
        out.__dict__.pop("_scales_cache", None)
        return out

    def dequantize(self, dtype=torch.float32) -> torch.Tensor:
        return nf4_dequantize(self.weight.data, self.weight.quant_state, self.weight.quant_state[1], dtype)

    # Opt-in, MI355X-specific: keep a decoded bf16 copy of the frozen weight next to the 4-bit codes.  288 GB of HBM hold
    # the bf16 copies of a 7B model (14 GB) with room to spare; the checkpoint and the optimizer state stay 4-bit / LoRA-only,
    # but every product becomes a plain library GEMM with no decode in the loop (2-5x at <= 2048 rows, where the fused
    # kernel's grid cannot fill 256 CUs, and no per-call decode above).  Dropped by load_dense() / merge().
    _dense_cache = None

    def cache_dense(self, enable: bool = True):
        self._dense_cache = self.dequantize(torch.bfloat16) if enable else None
        return self

    # the block scales travel with the module state (state_dict round trips of a quantised model)
    def get_extra_state(self):
        qs = self.weight.quant_state
        st = {"absmax": qs[0].detach().cpu(), "shape": tuple(qs[1]), "blocksize": qs[3], "quant_type": qs[5]}
        if qs[4] is not None:
            offset, (absmax2, code2) = qs[4]
            st["double_quant"] = {"offset": offset.detach().cpu(), "absmax2": absmax2.detach().cpu(), "code2": code2.detach().cpu()}
        return st

    def set_extra_state(self, state):
        qs = self.weight.quant_state
        dev = self.weight.device
        qs[0] = state["absmax"].to(dev)
        qs[1] = torch.Size(state["shape"])
        dq = state.get("double_quant")
        qs[4] = None if dq is None else [dq["offset"].to(dev), [dq["absmax2"].to(dev), dq["code2"].to(dev)]]
        self.weight.quant_state = qs
        self._dense_cache = None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return qlora_linear(x, self, None, None)


_announced = set()


def _announce_once(key: str, msg: str) -> None:
    """a route other than the hand-written one was taken for a reason the caller did not ask for: say so, once per reason"""
    if key not in _announced:
        _announced.add(key)
        logging.getLogger(__name__).warning(msg)


# ---------------------------------------------------------------------------------------------
# the fused op
# ---------------------------------------------------------------------------------------------
def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


# Above this many rows of x the frozen weight is decoded ONCE into dense bf16 (HIP kernel; kept resident across passes within
# RESIDENT_BYTES, else a per-stream scratch) and the product is the hand-written 256 x 256-tile GEMM of nf4_gemm.hip (route
# "gemm"; "library" = hipBLASLt for A/B): the 128-tile fused kernel re-decodes each W tile in every one of the M/128 workgroup
# rows and is vector-ALU-bound there (measured 0.54-0.72 PF/s against 1.1-1.5 PF/s dense at M = 8k..16k); below it the fused
# kernel wins on weight bytes (0.5 B/element instead of 2).
DENSE_M = int(os.environ.get("FASTMAX_NF4_DENSE_M", "2048"))
_dense_scratch = {}


def _dense_weight(wq, scales, N, K):
    """bf16 (N, K) view of a scratch buffer holding the decoded weight.  One buffer per (device, stream): valid until the next
    QLoRA call on that stream; work on another stream gets its own buffer instead of racing on this one."""
    key = (wq.device, torch.cuda.current_stream(wq.device).cuda_stream)
    buf = _dense_scratch.get(key)
    if buf is None or buf.numel() < N * K:
        buf = torch.empty(N * K, dtype=torch.bfloat16, device=wq.device)
        _dense_scratch[key] = buf
    with torch.cuda.device(wq.device):
        rc = _lib.lib().fastmax_hip_nf4_dequantize_s(wq.data_ptr(), scales.ref, buf.data_ptr(), N * K, _lib.BF16,
                                                     _stream(wq.device))
    _lib.check(rc, "fastmax_hip_nf4_dequantize_s")
    return buf[: N * K].view(N, K)


class _QLoRALinearFn(torch.autograd.Function):
    """y = x deq(W)^T + bias + ea eb^T.  Few rows: all in one HIP kernel (dx through the same dequant GEMM).  Many rows
    (M >= DENSE_M, bf16): HIP decode into the scratch + the tile GEMM.  d(ea), d(eb): the streaming rank-r kernels."""

    @staticmethod
    def forward(ctx, x2, ea, eb, wq, scales, bias, N, K, wdense=None, owner=None):
        M = x2.shape[0]
        ctx.scales = scales
        ctx.owner = owner
        dt = _lib.BF16 if x2.dtype == torch.bfloat16 else _lib.F32
        ctx.dense = dt == _lib.BF16 and (M >= DENSE_M or wdense is not None)
        ctx.wdense = wdense
        ctx.hip_gemm = ctx.dense and QLORA_ROUTE != "library" and N % 64 == 0 and K % 64 == 0 and \
            (ea is None or ea.shape[1] in (16, 32))
        if ctx.hip_gemm:
            # every product in libfastmax_hip.so: HIP decode into the scratch + the 256 x 256-tile GEMM (bias and the LoRA
            # branch fused as its last step)
            y = hip_gemm(x2, wdense if wdense is not None else _dense_weight(wq, scales, N, K), None, bias, ea, eb, N)
            ctx.save_for_backward(ea, eb, wq)
            ctx.dims = (M, N, K, dt)
            return y
        if ctx.dense:
            if QLORA_ROUTE != "library":
                _announce_once(f"dense-lib-{N % 64}-{K % 64}-{0 if ea is None else ea.shape[1]}",
                               f"QLoRA linear ({M} x {K}) -> {N}: the tile GEMM needs N and K to be multiples of 64 and a LoRA operand of "
                               "16 or 32 columns; this layer runs as HIP decode + library GEMM")
            y = x2 @ (wdense if wdense is not None else _dense_weight(wq, scales, N, K)).t()
            if ea is not None:
                y.addmm_(ea, eb.t())
            if bias is not None:
                y += bias.to(y.dtype)
            ctx.save_for_backward(ea, eb, wq)
            ctx.dims = (M, N, K, dt)
            return y
        y = torch.empty((M, N), dtype=x2.dtype, device=x2.device)
        with torch.cuda.device(x2.device):
            rc = _lib.lib().fastmax_hip_nf4_linear_forward_s(
                x2.data_ptr(), x2.stride(0), wq.data_ptr(), scales.ref,
                None if bias is None else bias.data_ptr(), None if ea is None else ea.data_ptr(),
                None if eb is None else eb.data_ptr(), y.data_ptr(), N, M, N, K, dt, _stream(x2.device))
        _lib.check(rc, "fastmax_hip_nf4_linear_forward_s")
        ctx.save_for_backward(ea, eb, wq)
        ctx.dims = (M, N, K, dt)
        return y

    @staticmethod
    def backward(ctx, dy):
        ea, eb, wq = ctx.saved_tensors
        scales = ctx.scales
        M, N, K, dt = ctx.dims
        dy = dy.contiguous()
        dx = d_ea = d_eb = None
        if ctx.needs_input_grad[0] and ctx.hip_gemm:
            wt = _resident_weight(ctx.owner, True) if M >= DENSE_M else None
            dx = hip_gemm(dy, wt if wt is not None else _dense_weight_t(wq, scales, N, K), None, None, None, None, K)
        elif ctx.needs_input_grad[0] and ctx.dense:
            dx = dy @ (ctx.wdense if ctx.wdense is not None else _dense_weight(wq, scales, N, K))
        elif ctx.needs_input_grad[0]:
            dx = torch.empty((M, K), dtype=dy.dtype, device=dy.device)
            with torch.cuda.device(dy.device):
                rc = _lib.lib().fastmax_hip_nf4_linear_backward_input_s(dy.data_ptr(), N, wq.data_ptr(), scales.ref,
                                                                        dx.data_ptr(), K, M, N, K, dt, _stream(dy.device))
            _lib.check(rc, "fastmax_hip_nf4_linear_backward_input_s")
        if ea is not None:
            dyb = dy.to(torch.bfloat16)
            RP = ea.shape[1]
            thin = (LORA_THIN and dyb.is_cuda and RP in (16, 32) and N % 64 == 0 and ea.dtype == torch.bfloat16 and
                    eb.dtype == torch.bfloat16 and dyb.stride(1) == 1 and (dyb.stride(0) * 2) % 16 == 0 and dyb.data_ptr() % 16 == 0)
            if thin:
                # the rank-r products of the backward pass as streaming HIP kernels (lora_thin.hip), one pass over dy each
                if ctx.needs_input_grad[1]:
                    d_ea, _ = lora_down(dyb, eb.t().contiguous(), want_t=False)          # (M, RP) = dy eb
                if ctx.needs_input_grad[2]:
                    MP = (M + 15) // 16 * 16
                    eat = torch.zeros((RP, MP), dtype=torch.bfloat16, device=dyb.device)
                    eat[:, :M] = ea.t()
                    d_eb = lora_tn(eat, dyb, dtype=torch.bfloat16, transpose=True)       # (N, RP) = dy^T ea
            else:
                if ctx.needs_input_grad[1]:
                    d_ea = dyb @ eb                   # shapes the thin kernels do not take: tensor ops
                if ctx.needs_input_grad[2]:
                    d_eb = dyb.t() @ ea
        return dx, d_ea, d_eb, None, None, None, None, None, None, None


# ---- rank-r products around the library GEMM (csrc/lora_thin.hip) ------------------------------------------------
LORA_THIN = os.environ.get("FASTMAX_LORA_THIN", "1") != "0"


def _pad_rank(r: int) -> int:
    return 16 if r <= 16 else 32


def lora_down(x: torch.Tensor, bt: torch.Tensor, want_t: bool = True, drop=None):
    """e (M, RP) = x (M, K) bt (RP, K)^T, and e^T (RP, roundup(M, 16)) zero padded; bf16.
    drop = (seed tensor on the device, p): x passes through the dropout mask of that seed first (scaled by 1 / (1 - p))."""
    M, K = x.shape
    RP = bt.shape[0]
    e = torch.empty((M, RP), dtype=torch.bfloat16, device=x.device)
    MP = (M + 15) // 16 * 16
    et = torch.empty((RP, MP), dtype=torch.bfloat16, device=x.device) if want_t else None
    with torch.cuda.device(x.device):
        if drop is None:
            rc = _lib.lib().fastmax_hip_lora_down(x.data_ptr(), x.stride(0), bt.data_ptr(), bt.stride(0), e.data_ptr(), RP,
                                                  None if et is None else et.data_ptr(), MP, M, K, RP, _stream(x.device))
        else:
            rc = _lib.lib().fastmax_hip_lora_down_dropout(x.data_ptr(), x.stride(0), bt.data_ptr(), bt.stride(0), e.data_ptr(), RP,
                                                          None if et is None else et.data_ptr(), MP, M, K, RP, drop[0].data_ptr(),
                                                          float(drop[1]), _stream(x.device))
    _lib.check(rc, "fastmax_hip_lora_down")
    return e, et


def new_dropout_seed(device) -> torch.Tensor:
    """one fresh 32-bit seed on the device, drawn from torch's device generator (so torch.manual_seed reproduces the masks and a
    captured HIP graph draws a new one on every replay)"""
    return torch.randint(0, 2 ** 31 - 1, (2,), dtype=torch.int32, device=device)


def dropout_mask(seed: torch.Tensor, M: int, K: int, p: float) -> torch.Tensor:
    """the (M, K) keep mask the rank-r kernels regenerate from `seed` (bool tensor; tests / inspection)"""
    mask = torch.empty((M, K), dtype=torch.uint8, device=seed.device)
    with torch.cuda.device(seed.device):
        rc = _lib.lib().fastmax_hip_lora_dropout_mask(mask.data_ptr(), M, K, seed.data_ptr(), float(p), _stream(seed.device))
    _lib.check(rc, "fastmax_hip_lora_dropout_mask")
    return mask.bool()


def dropout_scale(p: float) -> float:
    """1 / keep probability as the kernels apply it (the threshold is a 16-bit integer)"""
    t = min(65535, int(p * 65536.0 + 0.5))
    return 65536.0 / (65536 - t)


def lora_tn(et: torch.Tensor, x: torch.Tensor, R: int = None, dtype=torch.float32, transpose: bool = False, drop=None) -> torch.Tensor:
    """et[:R] (R, M) x (M, ncols) -> (R, ncols), or (ncols, R) when ``transpose``; float32 sums, result float32 or bf16.
    drop = (seed, p): x is replaced by dropout(x) of that seed."""
    M, ncols = x.shape
    RP = et.shape[0]
    R = RP if R is None else R
    L = _lib.lib()
    nb = L.fastmax_hip_lora_tn_workspace(M, ncols, RP)
    if nb < 0:
        raise NotImplementedError(f"lora_tn: unsupported shape M={M} ncols={ncols} RP={RP}")
    kdt = torch.bfloat16 if dtype == torch.bfloat16 else torch.float32
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
    out = torch.empty((ncols, R) if transpose else (R, ncols), dtype=kdt, device=x.device)
    with torch.cuda.device(x.device):
        if drop is None:
            rc = L.fastmax_hip_lora_tn(et.data_ptr(), et.stride(0), x.data_ptr(), x.stride(0), out.data_ptr(),
                                       _lib.BF16 if kdt == torch.bfloat16 else _lib.F32, int(transpose), R, ws.data_ptr(),
                                       M, ncols, RP, _stream(x.device))
        else:
            rc = L.fastmax_hip_lora_tn_dropout(et.data_ptr(), et.stride(0), x.data_ptr(), x.stride(0), out.data_ptr(),
                                               _lib.BF16 if kdt == torch.bfloat16 else _lib.F32, int(transpose), R, ws.data_ptr(),
                                               M, ncols, RP, drop[0].data_ptr(), float(drop[1]), _stream(x.device))
    _lib.check(rc, "fastmax_hip_lora_tn")
    return out if dtype == kdt else out.to(dtype)


def lora_up_(y: torch.Tensor, e: torch.Tensor, bn: torch.Tensor, bias=None, transposed: bool = False, drop=None) -> torch.Tensor:
    """y (M, N) += e (M, R) bn^T (+ bias), in place; bn is (N, R), or (R, N) with ``transposed``; bf16, bias float32.
    drop = (seed, p): the added product passes through the dropout mask of that seed (y has the shape of the dropped-out x)."""
    M, N = y.shape
    with torch.cuda.device(y.device):
        if drop is None:
            rc = _lib.lib().fastmax_hip_lora_up(y.data_ptr(), y.stride(0), e.data_ptr(), e.stride(0), bn.data_ptr(), bn.stride(0),
                                                int(transposed), None if bias is None else bias.data_ptr(), M, N, e.shape[1],
                                                _stream(y.device))
        else:
            rc = _lib.lib().fastmax_hip_lora_up_dropout(y.data_ptr(), y.stride(0), e.data_ptr(), e.stride(0), bn.data_ptr(), bn.stride(0),
                                                        int(transposed), None if bias is None else bias.data_ptr(), M, N, e.shape[1],
                                                        drop[0].data_ptr(), float(drop[1]), _stream(y.device))
    _lib.check(rc, "fastmax_hip_lora_up")
    return y


_KDT = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16}


class _ScatterRowsFn(torch.autograd.Function):
    """lora_B (n_rows, r) -> the (RP, N) bf16 operand  E^T[part r + j][n] = scaling B[row(part, n)][j]  of the LoRA branch
    (LoRAQKVLinear: the lora_ind / zero_pad scatter of lit_gpt/lora.py:263-342; LoRALinear: one part, identity map), one HIP
    launch each way instead of zeros + index_put + mul (+ transpose copies)."""

    @staticmethod
    def forward(ctx, B, rowmap, ind, part, scaling, N, RP):
        Bc = B.detach().contiguous()
        et = torch.empty((RP, N), dtype=torch.bfloat16, device=B.device)
        with torch.cuda.device(B.device):
            rc = _lib.lib().fastmax_hip_lora_scatter(Bc.data_ptr(), _KDT[Bc.dtype], Bc.shape[1], rowmap.data_ptr(), rowmap.shape[0],
                                                     float(scaling), et.data_ptr(), N, N, RP, _stream(B.device))
        _lib.check(rc, "fastmax_hip_lora_scatter")
        ctx.save_for_backward(ind, part)
        ctx.meta = (Bc.shape, Bc.dtype, float(scaling))
        return et

    @staticmethod
    def backward(ctx, d_et):
        ind, part = ctx.saved_tensors
        shape, dt, scaling = ctx.meta
        if d_et.dtype not in _KDT:
            d_et = d_et.float()
        d_et = d_et.contiguous()
        dB = torch.empty(shape, dtype=dt, device=d_et.device)
        with torch.cuda.device(d_et.device):
            rc = _lib.lib().fastmax_hip_lora_scatter_backward(d_et.data_ptr(), _KDT[d_et.dtype], d_et.stride(0), ind.data_ptr(),
                                                              part.data_ptr(), scaling, dB.data_ptr(), _KDT[dt], shape[0], shape[1],
                                                              _stream(d_et.device))
        _lib.check(rc, "fastmax_hip_lora_scatter_backward")
        return dB, None, None, None, None, None, None


# How the frozen product runs at training row counts (M >= DENSE_M, bf16):
#   "gemm"    (default) hand-written 256 x 256-tile GEMM (csrc/nf4_gemm.hip) on the weight decoded ONCE per call into a bf16
#             scratch by a HIP kernel, bias and the LoRA branch fused into the GEMM (one more 32-deep step); dx through the same
#             kernel on W^T (decoded transposed).  Measured (profiles/r02_qlora_gemm.md): 9-33 % faster than the library route
#   "fused"   the same kernel with the NF4 codes decoded INSIDE its loop (no scratch): wins only where M / 256 is small --
#             every 256-row block re-decodes the weight tile and the decode shares the vector ALU / LDS with the fragments
#   "library" decode once + hipBLASLt through torch.matmul + the rank-r streaming kernels (round 1's route; kept for A/B)
QLORA_ROUTE = os.environ.get("FASTMAX_QLORA_ROUTE", "gemm")


def hip_gemm(x2: torch.Tensor, w: torch.Tensor, scales, bias, ea, eb, N: int) -> torch.Tensor:
    """y (M, N) = x2 (M, K) w^T + bias + ea (M, RP) eb (N, RP)^T in libfastmax_hip.so; w: dense bf16 (N, K) when ``scales`` is
    None, else the packed NF4 codes of an (N, K) weight with their block scales (decoded inside the kernel's loop)."""
    M, K = x2.shape
    y = torch.empty((M, N), dtype=torch.bfloat16, device=x2.device)
    with torch.cuda.device(x2.device):
        rc = _lib.lib().fastmax_hip_qlora_gemm(x2.data_ptr(), x2.stride(0), w.data_ptr(), 0 if scales is None else 1,
                                               None if scales is None else scales.ref, None if bias is None else bias.data_ptr(),
                                               None if ea is None else ea.data_ptr(), None if eb is None else eb.data_ptr(),
                                               0 if ea is None else ea.shape[1], y.data_ptr(), y.stride(0), M, N, K,
                                               _stream(x2.device))
    _lib.check(rc, "fastmax_hip_qlora_gemm")
    return y


_dense_scratch_t = {}


# Decoded copies of FROZEN NF4 weights kept across the forward and backward passes of a step (and across steps): W (N, K) for the
# product and W^T (K, N) for dx, decoded once per layer by the HIP kernels instead of once per layer per pass (2 x ~25 us per
# linear per step at Llama-2-7B widths).  Kept per module up to a global budget -- the bf16 copies of a whole 7B model are 26 GB
# of the MI355X's 288 GB; beyond the budget a layer falls back to the per-call scratch decode.  Keyed by the identity of the code
# tensor, so quantize_base / merge / load / .to() start over.  FASTMAX_DENSE_RESIDENT_BYTES=0 turns it off.
RESIDENT_BYTES = int(os.environ.get("FASTMAX_DENSE_RESIDENT_BYTES", str(64 << 30)))
_resident = weakref.WeakKeyDictionary()           # NF4Linear -> {"tag": ..., "w": tensor | None, "wt": tensor | None}
_resident_used = [0]


def _resident_weight(base, transposed: bool):
    """the kept bf16 W (or W^T) of an NF4Linear, decoding it on first use; None when the budget is spent or residency is off"""
    if RESIDENT_BYTES <= 0 or base is None:
        return None
    wq = base.weight
    tag = (id(wq), wq.data_ptr(), wq._version, wq.device)
    ent = _resident.get(base)
    if ent is None or ent["tag"] != tag:
        if ent is not None:
            _resident_used[0] -= sum(t.numel() * 2 for t in (ent["w"], ent["wt"]) if t is not None)
        ent = {"tag": tag, "w": None, "wt": None}
        _resident[base] = ent
    key = "wt" if transposed else "w"
    if ent[key] is None:
        N, K = base.out_features, base.in_features
        if _resident_used[0] + N * K * 2 > RESIDENT_BYTES:
            return None
        scales = scales_of(base)
        src = _dense_weight_t(wq.data, scales, N, K) if transposed else _dense_weight(wq.data, scales, N, K)
        ent[key] = src.clone()
        _resident_used[0] += N * K * 2
    return ent[key]


def _dense_weight_t(wq, scales, N, K):
    """bf16 (K, N) = W^T decoded from the codes of W (N, K) into a per-(device, stream) scratch (the operand of dx = dy W)"""
    key = (wq.device, torch.cuda.current_stream(wq.device).cuda_stream)
    buf = _dense_scratch_t.get(key)
    if buf is None or buf.numel() < N * K:
        buf = torch.empty(N * K, dtype=torch.bfloat16, device=wq.device)
        _dense_scratch_t[key] = buf
    with torch.cuda.device(wq.device):
        rc = _lib.lib().fastmax_hip_nf4_dequantize_transposed(wq.data_ptr(), scales.ref, buf.data_ptr(), N, K, _stream(wq.device))
    _lib.check(rc, "fastmax_hip_nf4_dequantize_transposed")
    return buf[: N * K].view(K, N)


def hip_gemm_rope(x2, w, bias, ea, eb, N, cos32, sin32, B, T, G, qpk, hs, rope_n, tables16):
    """the qkv projection straight into q (B, G qpk, T, hs) and k, v (B, G, T, hs) with RoPE applied: one kernel
    (fastmax_hip_qlora_gemm_rope); x2: (B T, K) rows b * T + t, w: dense bf16 (N, K)"""
    M, K = x2.shape
    q = torch.empty((B, G * qpk, T, hs), dtype=torch.bfloat16, device=x2.device)
    k = torch.empty((B, G, T, hs), dtype=torch.bfloat16, device=x2.device)
    v = torch.empty((B, G, T, hs), dtype=torch.bfloat16, device=x2.device)
    with torch.cuda.device(x2.device):
        rc = _lib.lib().fastmax_hip_qlora_gemm_rope(x2.data_ptr(), x2.stride(0), w.data_ptr(), None if bias is None else bias.data_ptr(),
                                                    None if ea is None else ea.data_ptr(), None if eb is None else eb.data_ptr(),
                                                    0 if ea is None else ea.shape[1], cos32.data_ptr(), sin32.data_ptr(),
                                                    q.data_ptr(), k.data_ptr(), v.data_ptr(), M, N, K, T, G, qpk, hs, rope_n,
                                                    1 if tables16 else 0, _stream(x2.device))
    _lib.check(rc, "fastmax_hip_qlora_gemm_rope")
    return q, k, v


def gemm_rope_supported(N, K, M, T, G, qpk, hs, rope_n) -> bool:
    """shapes the fused qkv projection + RoPE epilogue takes (the 256 x 256-tile kernel: more than 128 tiles, whole heads
    per tile); FASTMAX_GEMM_ROPE=0 turns it off"""
    return (os.environ.get("FASTMAX_GEMM_ROPE", "1") != "0" and QLORA_ROUTE == "gemm" and N == G * (qpk + 2) * hs and 256 % hs == 0
            and rope_n % 16 == 0 and K % 64 == 0 and M % T == 0 and ((M + 255) // 256) * ((N + 255) // 256) > 128)


_frozen_wt = weakref.WeakKeyDictionary()


def _frozen_transpose(owner: nn.Module) -> torch.Tensor:
    """W^T (K, N) of a dense FROZEN base layer for the dx product, built once per layer (and again if the weight is written or
    replaced: the entry carries the parameter's identity, version counter and device) instead of once per layer per backward
    pass; it lives as long as the layer does.  A second N K 2 bytes per dense-base layer, on a 288 GB part."""
    w = owner.weight
    tag = (id(w), w._version, w.data_ptr(), w.device)
    hit = _frozen_wt.get(owner)
    if hit is None or hit[0] != tag:
        hit = (tag, w.detach().t().contiguous())
        _frozen_wt[owner] = hit
    return hit[1]


class _QLoRAGemmFn(torch.autograd.Function):
    """y = x deq(W)^T + bias + (x A^T) eb^T with every product in libfastmax_hip.so (lit_gpt/lora.py:170-177, 419-433 and
    their autograd mirror, without dropout): the frozen product and dx by the 256 x 256-tile GEMM with the LoRA branch as
    its last step, x A^T / dy eb by lora_down, dA / dB by lora_tn."""

    @staticmethod
    def forward(ctx, x2, A, ebt, wq, scales, bias, N, K, wdense, fused, rope=None, owner=None, drop_p=0.0):
        """rope = (cos32, sin32, B, T, G, qpk, hs, rope_n, tables16, expand): the product is an attention sub-layer's qkv
        projection and leaves the kernel as (q, k, v) -- de-interleaved and rotated, in the layout of ops.RopeQKVSplit's
        `expand` mode 0 (k, v at their G heads; also mode 1 when qpk == 1), 3 or 4 (stride-0 group views) -- instead of y"""
        R, RP = A.shape[0], ebt.shape[0]
        ctx.scales = scales
        if R == RP:
            abt = A.detach().to(torch.bfloat16).contiguous()
        else:
            abt = torch.zeros((RP, K), dtype=torch.bfloat16, device=x2.device)
            abt[:R] = A.detach()
        # LoRA dropout (lit_gpt/lora.py:175, 422): the branch sees dropout(x); the mask is a function of a per-call seed and is
        # regenerated by the backward kernels (lora_thin.hip), never stored
        ctx.drop = (new_dropout_seed(x2.device), float(drop_p)) if drop_p > 0.0 else None
        ea, eat = lora_down(x2, abt, drop=ctx.drop)
        eb = ebt.t().contiguous()                                  # (N, RP): the B-side operand of the GEMM's last step
        ctx.save_for_backward(x2, eat, abt, ebt, wq)
        ctx.dims = (N, K, R, A.dtype)
        ctx.rope = rope
        ctx.dense_base = wdense if wq is None else None           # a dense frozen base: its own weight serves dx
        ctx.dense_owner = owner if wq is None else None           # ... and its layer keeps the transposed copy
        ctx.nf4_owner = owner if wq is not None else None         # an NF4 layer: its resident W^T (if any) serves dx
        if rope is not None:
            cos32, sin32, B, T, G, qpk, hs, rope_n, tables16, expand = rope
            w = wdense if wdense is not None else _dense_weight(wq, scales, N, K)
            q, k, v = hip_gemm_rope(x2, w, bias, ea, eb, N, cos32, sin32, B, T, G, qpk, hs, rope_n, tables16)
            if expand in (3, 4):                                      # ops.RopeQKVSplit's group views: nothing is copied
                kv = lambda t: t.view(B * G, 1, T, hs).expand(B * G, qpk, T, hs)
                return q.view(B * G, qpk, T, hs), (kv(k) if expand == 3 else k), kv(v)
            return q, k, v
        if fused and wdense is None:
            y = hip_gemm(x2, wq, scales, bias, ea, eb, N)
        else:
            y = hip_gemm(x2, wdense if wdense is not None else _dense_weight(wq, scales, N, K), None, bias, ea, eb, N)
        return y

    @staticmethod
    def backward(ctx, dy, *more):
        x2, eat, abt, ebt, wq = ctx.saved_tensors
        N, K, R, a_dt = ctx.dims
        if ctx.rope is not None:
            # gradients of q (B, G qpk, T, hs), k, v (B, G, T, hs): inverse rotation + re-interleave in one pass, then as before
            from . import ops
            cos32, sin32, B, T, G, qpk, hs, rope_n, _, expand = ctx.rope
            # modes 3 / 4: the gradients of the stride-0 views arrive dense, one per query head, and the pass sums a group
            # while it reads them -- exactly ops.RopeQKVSplit.backward
            dy = ops.rope_qkv_backward(dy, more[0], more[1], cos32, sin32, B, T, G, qpk, hs, rope_n,
                                       {0: 0, 1: 0, 3: 1, 4: 2}[expand]).view(B * T, N)
        dy = dy.contiguous()
        dx = dA = d_ebt = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            d_ea, d_eat = lora_down(dy, ebt)
        if ctx.needs_input_grad[0]:
            # dx = dy W + d_ea abt: the GEMM over n with W^T as its weight operand and the LoRA step (d_ea, abt^T)
            if ctx.dense_base is None:
                wt = _resident_weight(ctx.nf4_owner, True)
                if wt is None:
                    wt = _dense_weight_t(wq, ctx.scales, N, K)
            else:
                wt = _frozen_transpose(ctx.dense_owner) if ctx.dense_owner is not None else ctx.dense_base.t().contiguous()
            if ctx.drop is None:
                dx = hip_gemm(dy, wt, None, None, d_ea, abt.t().contiguous(), K)
            else:
                # the branch's share of dx passes through the same mask: the frozen product alone, then one masked rank update
                dx = lora_up_(hip_gemm(dy, wt, None, None, None, None, K), d_ea, abt, transposed=True, drop=ctx.drop)
        if ctx.needs_input_grad[1]:
            dA = lora_tn(d_eat, x2, R, a_dt, drop=ctx.drop)
        if ctx.needs_input_grad[2]:
            d_ebt = lora_tn(eat, dy, dtype=torch.bfloat16)
        return dx, dA, d_ebt, None, None, None, None, None, None, None, None, None, None


class _QLoRAThinFn(torch.autograd.Function):
    """The many-rows route with the LoRA branch in libfastmax_hip.so: y = x deq(W)^T + bias + (x A^T) eb^T.
    Base products are library GEMMs on the decoded (or cached) weight; the rank-r products are lora_down / lora_tn / lora_up,
    one streaming pass each (lit_gpt/lora.py:170-177, 419-433 and their autograd mirror, without dropout)."""

    @staticmethod
    def forward(ctx, x2, A, ebt, wq, scales, bias, N, K, wdense):
        R, RP = A.shape[0], ebt.shape[0]
        ctx.scales = scales
        if R == RP:                                   # no padding needed: use A as it is
            abt = A.detach().to(torch.bfloat16).contiguous()
        else:
            abt = torch.zeros((RP, K), dtype=torch.bfloat16, device=x2.device)
            abt[:R] = A.detach()
        y = x2 @ (wdense if wdense is not None else _dense_weight(wq, scales, N, K)).t()
        ea, eat = lora_down(x2, abt)
        lora_up_(y, ea, ebt, bias, transposed=True)
        ctx.save_for_backward(x2, eat, abt, ebt, wq)
        ctx.wdense = wdense
        ctx.dims = (N, K, R, A.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, eat, abt, ebt, wq = ctx.saved_tensors
        scales = ctx.scales
        N, K, R, a_dt = ctx.dims
        dy = dy.contiguous()
        dx = dA = d_ebt = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            d_ea, d_eat = lora_down(dy, ebt)
        if ctx.needs_input_grad[0]:
            dx = dy @ (ctx.wdense if ctx.wdense is not None else _dense_weight(wq, scales, N, K))
            lora_up_(dx, d_ea, abt, transposed=True)
        if ctx.needs_input_grad[1]:
            dA = lora_tn(d_eat, x2, R, a_dt)
        if ctx.needs_input_grad[2]:
            d_ebt = lora_tn(eat, dy, dtype=torch.bfloat16)
        return dx, dA, d_ebt, None, None, None, None, None, None


def thin_route(x: torch.Tensor, base: "NF4Linear") -> bool:
    """Does this input take the library-GEMM route with the HIP rank-r kernels?"""
    if not LORA_THIN or x.device.type != "cuda" or x.dtype != torch.bfloat16:
        return False
    M = x.numel() // x.shape[-1]
    return M >= DENSE_M or base._dense_cache is not None


def qlora_linear_thin(x, base: "NF4Linear", A, ebt, rope=None, drop_p: float = 0.0):
    """x: (..., K) bf16 device tensor; A: (r, K); ebt: (RP, N) bf16 operand of the branch (scaling applied, rank zero padded);
    drop_p: LoRA dropout probability in effect (0 outside training)."""
    N, K = base.out_features, base.in_features
    if K % 128 or N % 64:
        raise NotImplementedError(f"fused NF4 linear needs in_features % 128 == 0 and out_features % 64 == 0, got {K}, {N}")
    x2 = x.reshape(-1, K)
    if x2.stride(1) != 1 or (x2.stride(0) * x2.element_size()) % 16 or x2.data_ptr() % 16:
        x2 = x2.contiguous()
    bias = None if base.bias is None else base.bias.data
    if bias is not None and bias.dtype != torch.float32:
        bias = bias.float()
    if not isinstance(base, NF4Linear):
        # a dense bf16 frozen base (LoRA without quantisation, lit_gpt/lora.py:170-177): the same tile GEMM on the weight itself
        y = _QLoRAGemmFn.apply(x2, A, ebt, None, None, bias, N, K, base.weight.data, False, rope, base, drop_p)
        return y if rope is not None else y.reshape(*x.shape[:-1], N)
    scales = scales_of(base)
    if QLORA_ROUTE != "library" and N % 64 == 0 and K % 64 == 0:
        fused = QLORA_ROUTE == "fused" and rope is None
        wd = base._dense_cache
        if wd is None and not fused:
            wd = _resident_weight(base, False)                       # decoded once per layer, not once per pass
        if rope is not None:
            return _QLoRAGemmFn.apply(x2, A, ebt, base.weight.data, scales, bias, N, K, wd, False, rope, base, drop_p)
        y = _QLoRAGemmFn.apply(x2, A, ebt, base.weight.data, scales, bias, N, K, wd, fused, None, base, drop_p)
    else:
        if drop_p > 0.0:
            raise NotImplementedError("LoRA dropout on the many-rows route needs FASTMAX_QLORA_ROUTE=gemm (the default) or fused")
        y = _QLoRAThinFn.apply(x2, A, ebt, base.weight.data, scales, bias, N, K, base._dense_cache)
    return y.reshape(*x.shape[:-1], N)


def qlora_linear(x, base: NF4Linear, ea, eb):
    """x: (..., K) device tensor (bf16 or f32). ea: (M, r_tot) or None, eb: (N, r_tot) or None."""
    if x.device.type != "cuda":
        raise RuntimeError("the NF4 + LoRA linear runs on an MI355X only; there is no CPU fallback")
    N, K = base.out_features, base.in_features
    if K % 128 or N % 64:
        raise NotImplementedError(f"fused NF4 linear needs in_features % 128 == 0 and out_features % 64 == 0, got {K}, {N}")
    cdt = x.dtype if x.dtype in (torch.bfloat16, torch.float32) else torch.bfloat16
    x2 = x.reshape(-1, K).to(cdt)
    if x2.stride(1) != 1 or (x2.stride(0) * x2.element_size()) % 16 or x2.data_ptr() % 16:
        x2 = x2.contiguous()
    if ea is not None:
        r = ea.shape[-1]
        library_route = cdt == torch.bfloat16 and (x2.shape[0] >= DENSE_M or base._dense_cache is not None)
        if library_route and QLORA_ROUTE != "library" and r <= 32:
            # decode-once route on the hand-written GEMM: the branch is its last step, rank padded to 16 or 32
            rp = _pad_rank(r)
            ea = F.pad(ea.reshape(-1, r).to(torch.bfloat16), (0, rp - r)).contiguous()
            eb = F.pad(eb.to(torch.bfloat16), (0, rp - r)).contiguous()
        elif library_route:
            # decode-once / cached route: the LoRA branch is a plain addmm -- no padding to the fused kernel's 32-wide k-step
            ea, eb = ea.reshape(-1, r).to(torch.bfloat16), eb.to(torch.bfloat16)
        else:
            ea = F.pad(ea.reshape(-1, r).to(torch.bfloat16), (0, RANK_PAD - r)).contiguous()
            eb = F.pad(eb.to(torch.bfloat16), (0, RANK_PAD - r)).contiguous()
    bias = None if base.bias is None else base.bias.data
    if bias is not None and bias.dtype != torch.float32:
        bias = bias.float()
    wd = base._dense_cache
    if wd is None and cdt == torch.bfloat16 and x2.shape[0] >= DENSE_M:
        wd = _resident_weight(base, False)                       # training rows: the decoded weight stays (budgeted)
    y = _QLoRALinearFn.apply(x2, ea, eb, base.weight.data, scales_of(base), bias, N, K, wd, base)
    return y.reshape(*x.shape[:-1], N).to(x.dtype)


# ---------------------------------------------------------------------------------------------
# LoRA layers (reference interface)
# ---------------------------------------------------------------------------------------------
class LoRALayer(nn.Module):
    def __init__(self, r: int, lora_alpha: int, lora_dropout: float):
        super().__init__()
        assert r >= 0
        self.r, self.lora_alpha = r, lora_alpha
        self.lora_dropout = nn.Dropout(p=lora_dropout) if lora_dropout > 0.0 else (lambda x: x)
        self.merged = False


class LoRALinear(LoRALayer):
    """lit_gpt/lora.py:88-177."""

    def __init__(self, in_features: int, out_features: int, r: int = 0, lora_alpha: int = 1, lora_dropout: float = 0.0,
                 **kwargs: Any):
        super().__init__(r=r, lora_alpha=lora_alpha, lora_dropout=lora_dropout)
        self.linear = torch.nn.Linear(in_features, out_features, **kwargs)
        if r > 0:
            self.lora_A = nn.Parameter(torch.zeros((r, in_features)))
            self.lora_B = nn.Parameter(torch.zeros((out_features, r)))
            self.scaling = self.lora_alpha / self.r
            self.reset_parameters()

    def reset_parameters(self) -> None:
        if hasattr(self, "lora_A"):
            nn.init.kaiming_uniform_(self.lora_A, a=math.sqrt(5))
            nn.init.zeros_(self.lora_B)

    def quantize_base(self, double_quant: bool = False) -> "LoRALinear":
        """Swap the dense frozen layer for its NF4 version (what the bnb plugin does at construction); ``double_quant``:
        the "bnb.nf4-dq" mode of finetune/lora.py:38 (block scales stored in 8 bits)."""
        if not isinstance(self.linear, NF4Linear):
            self.linear = NF4Linear.from_linear(self.linear, double_quant=double_quant)
        return self

    def get_lora_AB(self) -> torch.Tensor:
        return (self.lora_B @ self.lora_A) * self.scaling

    def _dense_rows(self) -> torch.Tensor:
        """(out_features, r_total) matrix E with  lora(x) = (dropout(x) A^T) E^T * scaling."""
        return self.lora_B

    def _scatter_maps(self):
        """(rowmap (n_parts, N), ind (n_rows,), part (n_rows,)) int32: which lora_B row feeds output column n in each rank block"""
        N = self.linear.out_features
        ind = torch.arange(N, dtype=torch.int32)
        return ind[None, :].clone(), ind, torch.zeros(N, dtype=torch.int32)

    def _dense_rows_t(self) -> torch.Tensor:
        """(RP, out_features) bf16: scaling * _dense_rows()^T with the rank zero padded to 16 / 32 -- one HIP launch"""
        dev = self.lora_B.device
        maps = getattr(self, "_maps_cache", None)
        if maps is None or maps[0].device != dev:
            maps = tuple(t.to(dev).contiguous() for t in self._scatter_maps())
            self._maps_cache = maps
        return _ScatterRowsFn.apply(self.lora_B, maps[0], maps[1], maps[2], self.scaling, self.linear.out_features,
                                    _pad_rank(self.lora_A.shape[0]))

    def merge(self) -> None:
        """W <- W + dW (lora.py:142-168); the 4-bit branch dequantises, adds and requantises."""
        if self.r > 0 and not self.merged:
            lora_data = self.get_lora_AB()
            if isinstance(self.linear, NF4Linear):
                w = self.linear.dequantize(torch.float32) + lora_data.float().to(self.linear.weight.device)
                self.linear.load_dense(w.to(self.linear.weight.quant_state[2]))      # keeps nf4 / nf4-dq as it was
            elif self.linear.weight.data.dtype == lora_data.dtype:
                self.linear.weight.data += lora_data
            else:
                raise NotImplementedError(f"Cannot merge the pretrained weights of type {self.linear.weight.data.dtype}"
                                          f" and LoRA weights of type {lora_data.dtype}")
            self.merged = True

    def _lora_enabled(self) -> bool:
        return self.r > 0 and not self.merged

    def _drop_p(self) -> float:
        """LoRA dropout probability in effect for this call (lit_gpt/lora.py:79-83: nn.Dropout in training mode, else identity)"""
        d = self.lora_dropout
        return float(d.p) if isinstance(d, nn.Dropout) and self.training else 0.0

    def _hand_written_dropout_ok(self) -> bool:
        """dropout inside the rank-r kernels exists on the tile-GEMM route (FASTMAX_QLORA_ROUTE=gemm / fused)"""
        return self._drop_p() == 0.0 or QLORA_ROUTE != "library"

    def rope_fusable(self, x: torch.Tensor) -> bool:
        """would forward(x, rope=...) take the one-kernel route (qkv projection + de-interleave + RoPE)?"""
        if not (self._lora_enabled() and self.lora_A.shape[0] <= RANK_PAD):
            return False
        if not isinstance(self.linear, NF4Linear):
            return self._dense_base_on_tile_gemm(x)
        return thin_route(x, self.linear) and QLORA_ROUTE == "gemm"

    def forward(self, x: torch.Tensor, rope=None) -> torch.Tensor:
        if not self._lora_enabled():
            return self.linear(x)
        if isinstance(self.linear, NF4Linear) and self.lora_A.shape[0] <= RANK_PAD:
            if thin_route(x, self.linear) and self._hand_written_dropout_ok():
                return qlora_linear_thin(x, self.linear, self.lora_A, self._dense_rows_t(), rope, self._drop_p())
            ea = F.linear(self.lora_dropout(x), self.lora_A.to(x.dtype))
            return qlora_linear(x, self.linear, ea, self._dense_rows() * self.scaling)
        if self._dense_base_on_tile_gemm(x):
            return qlora_linear_thin(x, self.linear, self.lora_A, self._dense_rows_t(), rope, self._drop_p())
        pretrained = self.linear(x)
        lora = (self.lora_dropout(x) @ self.lora_A.transpose(0, 1).to(x.dtype)) @ self._dense_rows().transpose(0, 1).to(x.dtype)
        return pretrained + lora * self.scaling

    def _dense_base_on_tile_gemm(self, x: torch.Tensor) -> bool:
        """a dense (unquantised) frozen bf16 base at training row counts: the hand-written tile GEMM with the LoRA branch and
        the bias fused, like the NF4 route after its decode (FASTMAX_DENSE_LORA_GEMM=0: tensor ops, as the reference)"""
        lin = self.linear
        if isinstance(lin, NF4Linear) or not isinstance(lin, nn.Linear) or os.environ.get("FASTMAX_DENSE_LORA_GEMM", "1") == "0":
            return False
        M = x.numel() // x.shape[-1]
        w = lin.weight
        # the tile GEMM's autograd function treats the base (weight AND bias) as frozen data: a trainable bias
        # (mark_only_lora_as_trainable(bias="all" / "lora_only"), lit_gpt/lora.py:436-461) stays on the tensor-op route
        frozen_bias = lin.bias is None or not lin.bias.requires_grad
        return (LORA_THIN and QLORA_ROUTE == "gemm" and x.device.type == "cuda" and x.dtype == torch.bfloat16
                and w.dtype == torch.bfloat16 and not w.requires_grad and frozen_bias and w.is_contiguous() and M >= DENSE_M
                and self.lora_A.shape[0] <= RANK_PAD and lin.in_features % 128 == 0 and lin.out_features % 64 == 0)


class LoRAQKVLinear(LoRALinear):
    """lit_gpt/lora.py:180-433: LoRA on the fused, GQA-interleaved QKV projection."""

    def __init__(self, in_features: int, out_features: int, n_head: int, n_query_groups: int, r: int = 0,
                 lora_alpha: int = 1, lora_dropout: float = 0.0,
                 enable_lora: Union[bool, Tuple[bool, bool, bool]] = False, **kwargs: Any):
        super(LoRALinear, self).__init__(r=r, lora_alpha=lora_alpha, lora_dropout=lora_dropout)
        self.linear = torch.nn.Linear(in_features, out_features, **kwargs)
        self.n_head, self.n_query_groups = n_head, n_query_groups
        self.enable_lora = list(enable_lora) if not isinstance(enable_lora, bool) else [enable_lora] * 3
        if len(self.enable_lora) != 3:
            raise ValueError("enable_lora is one flag, or one flag each for q, k and v")
        if r > 0 and any(self.enable_lora):
            # The fused projection's columns come in n_query_groups runs of (q_per_kv query heads, one key head, one value
            # head), head_size columns each (lit_gpt/model.py:397-403).  slot[c] = position of column c's head in its run.
            q_per_kv = n_head // n_query_groups
            head_size = out_features // (n_query_groups * (q_per_kv + 2))
            slot = (torch.arange(out_features) // head_size) % (q_per_kv + 2)
            owner = torch.where(slot < q_per_kv, 0, slot - q_per_kv + 1)          # 0 = q, 1 = k, 2 = v for every column
            cols = [torch.nonzero(owner == part).flatten() for part in range(3) if self.enable_lora[part]]
            # names the reference's consumers read (lora.py:240-278): rows of lora_B per enabled part, and their columns
            self.kv_embd_size = n_query_groups * head_size
            self.qkv_shapes = [int(c.numel()) for c in cols]
            self.lora_ind = torch.cat(cols).tolist()
            self.lora_A = nn.Parameter(torch.zeros((r * len(cols), in_features)))
            self.lora_B = nn.Parameter(torch.zeros(sum(self.qkv_shapes), r))
            self.scaling = self.lora_alpha / self.r
            # rank block of every lora_B row (rows are ordered part by part): row i multiplies after_A[:, _cols[i]]
            block = torch.cat([torch.full((n,), j) for j, n in enumerate(self.qkv_shapes)])
            self.register_buffer("_ind", torch.cat(cols), persistent=False)
            self.register_buffer("_cols", block[:, None] * r + torch.arange(r)[None, :], persistent=False)
            self.reset_parameters()

    def zero_pad(self, x: torch.Tensor) -> torch.Tensor:
        """Place the enabled parts' columns at their positions in the full interleaved QKV width, zeros elsewhere
        (lora.py:281-342).  Activations (..., sum(qkv_shapes)) are widened along the last dimension; the 2-d product
        B A of get_lora_AB (sum(qkv_shapes), in_features) along its first (it becomes the weight update's rows)."""
        if all(self.enable_lora):
            return x
        dim = 0 if x.dim() == 2 else x.dim() - 1
        shape = list(x.shape)
        shape[dim] = self.linear.out_features
        return x.new_zeros(shape).index_copy_(dim, self._ind.to(x.device), x)

    def conv1d(self, input: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
        """Channels-first block-diagonal product (lora.py:344-377): input (B, r n_enabled, T), weight (sum(qkv_shapes), r, 1)
        -> (B, sum(qkv_shapes), T); part j's rows of `weight` see only rank block j of `input`."""
        w = weight.squeeze(-1)
        out, row = [], 0
        for j, n in enumerate(self.qkv_shapes):
            out.append(torch.matmul(w[row:row + n], input[:, j * self.r:(j + 1) * self.r]))
            row += n
        return torch.cat(out, dim=1)

    def get_lora_AB(self) -> torch.Tensor:
        lora = self.conv1d(self.lora_A.data.unsqueeze(0), self.lora_B.data.unsqueeze(-1)).squeeze(0)
        return self.zero_pad(lora * self.scaling)

    def _dense_rows(self) -> torch.Tensor:
        """(out_features, r*n_enabled): lora_B scattered to its output columns and rank block, zeros elsewhere,
        so that zero_pad(conv1d(after_A, B)) == after_A @ E^T   (lora.py:426-432 as one small dense operand)."""
        E = self.lora_B.new_zeros((self.linear.out_features, self.lora_A.shape[0]))
        # zero_pad is the identity when all three parts are enabled (lora.py:317-318): the rows then sit at
        # their own index, NOT at lora_ind -- kept as the reference has it
        ind = torch.arange(self.linear.out_features, device=E.device) if all(self.enable_lora) else self._ind.to(E.device)
        rows = ind[:, None].expand(-1, self.r)
        return E.index_put((rows, self._cols.to(E.device)), self.lora_B)

    def _scatter_maps(self):
        N = self.linear.out_features
        # rows sit at their own index when all three parts are enabled (the zero_pad identity quirk, see _dense_rows)
        ind = torch.arange(N) if all(self.enable_lora) else self._ind.cpu()
        part = (self._cols[:, 0] // self.r).cpu()
        rowmap = torch.full((len(self.qkv_shapes), N), -1, dtype=torch.int32)
        rowmap[part, ind] = torch.arange(ind.numel(), dtype=torch.int32)
        return rowmap, ind.to(torch.int32), part.to(torch.int32)

    def merge(self) -> None:
        if self.r > 0 and any(self.enable_lora) and not self.merged:
            super().merge()

    def _lora_enabled(self) -> bool:
        return self.r > 0 and any(self.enable_lora) and not self.merged


def mark_only_lora_as_trainable(model: nn.Module, bias: str = "none") -> None:
    """lit_gpt/lora.py:436-461: everything that is not a LoRA matrix is frozen; `bias` then thaws bias vectors again --
    "none": no bias trains, "all": every parameter with "bias" in its name, "lora_only": the `bias` attribute of LoRA layers
    that have one.  Anything else raises NotImplementedError, as the reference does."""
    if bias not in ("none", "all", "lora_only"):
        raise NotImplementedError(f"bias={bias!r}")
    for name, p in model.named_parameters():
        if "lora_" not in name:
            p.requires_grad = bias == "all" and "bias" in name
    if bias == "lora_only":
        for m in model.modules():
            own = getattr(m, "bias", None) if isinstance(m, LoRALayer) else None
            if own is not None:
                own.requires_grad = True


def lora_filter(key: str, value: Any) -> bool:
    """lit_gpt/lora.py:469-470."""
    return "lora_" in key


def enable_gemm_tuning(filename: str = None, tune: bool = True) -> None:
    """Opt-in: let PyTorch's TunableOp pick the hipBLASLt solution for the dense GEMMs of the many-rows route (the decoded
    frozen weight times x / dy).  The library's default heuristic is not the fastest solution at these shapes: measured on the
    TinyLlama sub-layer (16384 rows), 1.37 -> 1.25-1.30 ms forward+backward.  The first call of every new shape is timed
    over the candidate solutions (seconds); `filename` keeps the choices across runs."""
    torch.cuda.tunable.enable(True)
    torch.cuda.tunable.tuning_enable(bool(tune))
    if filename:
        torch.cuda.tunable.set_filename(filename)


def cache_dense_weights(model: nn.Module, enable: bool = True) -> int:
    """Opt-in bf16 copies of every NF4Linear weight in `model` (see NF4Linear.cache_dense).  Returns the bytes held."""
    held = 0
    for m in model.modules():
        if isinstance(m, NF4Linear):
            m.cache_dense(enable)
            if m._dense_cache is not None:
                held += m._dense_cache.numel() * 2
    return held


def merge_lora_weights(model: nn.Module) -> None:
    """lit_gpt/lora.py:473-477: merge every LoRA layer in place (4-bit bases: dequantise + add + requantise)."""
    for module in model.modules():
        if isinstance(module, LoRALinear):
            module.merge()


def save_lora_checkpoint(model: nn.Module, file_path) -> None:
    """finetune/lora.py:341-343: only the adapter weights go to disk (`filter={"model": lora_filter}`)."""
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items() if lora_filter(k, v)}
    torch.save({"model": sd}, str(file_path))


def load_lora_checkpoint(model: nn.Module, file_path) -> None:
    """scripts/merge_lora.py:70-73: adapter weights on top of the already-loaded base model (strict=False)."""
    ckpt = torch.load(str(file_path), map_location="cpu", weights_only=True)
    sd = ckpt.get("model", ckpt)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    if unexpected:
        raise KeyError(f"unexpected keys in LoRA checkpoint: {unexpected}")


def merged_state_dict(model: nn.Module, dtype=None) -> dict:
    """scripts/merge_lora.py:76-82: after merge_lora_weights, drop the LoRA parameters and the `linear.` level so the
    result loads into the plain (non-LoRA) model; 4-bit bases are written dense."""
    out = {}
    for name, module in model.named_modules():
        if isinstance(module, NF4Linear):
            w = module.dequantize(dtype or module.weight.quant_state[2])
            out[(name + ".weight").replace("linear.", "")] = w.detach().cpu()
            if module.bias is not None:
                out[(name + ".bias").replace("linear.", "")] = module.bias.detach().to(w.dtype).cpu()
    quant_prefixes = tuple(n + "." for n, m in model.named_modules() if isinstance(m, NF4Linear))
    for k, v in model.state_dict().items():
        if lora_filter(k, v) or k.startswith(quant_prefixes) or k.endswith("_extra_state"):
            continue
        out[k.replace("linear.", "")] = v.detach().cpu()
    return out
