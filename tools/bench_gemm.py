#!/usr/bin/env python3
"""The hand-written QLoRA GEMM (csrc/nf4_gemm.hip) against the library: correctness and time at training shapes.
usage: python tools/bench_gemm.py [M N K ...triples]"""
import ctypes
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fastmax_experiments_amd import _lib, lora


def gemm(x, w, scales=None, bias=None, ea=None, eb=None):
    M, K = x.shape
    N = w.shape[0] if scales is None else scales_shape[0]
    y = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    rc = _lib.lib().fastmax_hip_qlora_gemm(x.data_ptr(), x.stride(0), w.data_ptr(), 0 if scales is None else 1,
                                           None if scales is None else scales.ref, None if bias is None else bias.data_ptr(),
                                           None if ea is None else ea.data_ptr(), None if eb is None else eb.data_ptr(),
                                           0 if ea is None else ea.shape[1], y.data_ptr(), y.stride(0), M, N, K,
                                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "fastmax_hip_qlora_gemm")
    return y


def timeit(fn, iters=20, rounds=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters)
    return statistics.median(ts)


def main():
    global scales_shape
    args = [int(a) for a in sys.argv[1:]]
    shapes = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [(16384, 2560, 2048), (16384, 4096, 4096), (4096, 4096, 4096), (16384, 2048, 2048)]
    print("| M | N | K | kernel | ms | TFLOP/s | max rel err |")
    print("|---|---|---|---|---|---|---|")
    for M, N, K in shapes:
        g = torch.Generator(device="cuda").manual_seed(M + N + K)
        x = (torch.randn(M, K, device="cuda", generator=g)).to(torch.bfloat16)
        wf = torch.randn(N, K, device="cuda", generator=g) * 0.05
        lin = torch.nn.Linear(K, N, bias=True, device="cuda")
        lin.weight.data.copy_(wf)
        q = lora.NF4Linear.from_linear(lin)
        scales = lora.NF4Scales(q.weight.quant_state)
        scales_shape = (N, K)
        wd = q.dequantize(torch.bfloat16)
        ea = (torch.randn(M, 16, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
        eb = (torch.randn(N, 16, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
        ref = (x.float() @ wd.float().T + q.bias.float() + ea.float() @ eb.float().T)
        flops = 2.0 * M * N * K

        def row(name, fn, check=None):
            ms = timeit(fn)
            err = "" if check is None else f"{float((check().float() - ref).abs().max() / ref.abs().max()):.2e}"
            print(f"| {M} | {N} | {K} | {name} | {ms:.3f} | {flops / ms / 1e9:.0f} | {err} |", flush=True)

        row("library: x @ Wd^T (weight already dense)", lambda: x @ wd.t())
        row("library route: decode once + x @ Wd^T + addmm(ea, eb) + bias",
            lambda: (x @ lora._dense_weight(q.weight.data, scales, N, K).t()).addmm_(ea, eb.t()).add_(q.bias.to(torch.bfloat16)))
        for sched, label in ((0, "default loop (copies by one wave of a SIMD pair, fragments read under the MFMAs)"),
                             (14, "plain loop (every wave: copies, reads, MFMAs)"), (12, "128 x 256 tiles"), (13, "256 x 256 tiles forced")):
            _lib.check(_lib.lib().fastmax_hip_tune(b"gemm_sched", sched), "tune")
            row(f"hand-written, dense bf16 W, {label} (+ bias + LoRA step)", lambda: gemm(x, wd, None, q.bias, ea, eb),
                lambda: gemm(x, wd, None, q.bias, ea, eb))
        for sched, label in ((14, "plain loop, decode after the MFMAs"), (0, "default: fragments read under the MFMAs")):
            _lib.check(_lib.lib().fastmax_hip_tune(b"gemm_sched", sched), "tune")
            row(f"hand-written, NF4 decoded in the loop, {label} (+ bias + LoRA step)",
                lambda: gemm(x, q.weight.data, scales, q.bias, ea, eb), lambda: gemm(x, q.weight.data, scales, q.bias, ea, eb))
        _lib.check(_lib.lib().fastmax_hip_tune(b"gemm_sched", 0), "tune")
        row("hand-written pair (sched 0): HIP decode to bf16 scratch + dense-W kernel (+ bias + LoRA step)",
            lambda: gemm(x, lora._dense_weight(q.weight.data, scales, N, K), None, q.bias, ea, eb))


if __name__ == "__main__":
    main()
