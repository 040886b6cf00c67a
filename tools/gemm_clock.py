"""In-kernel clock and cycles per K step of the hand-written GEMM (diagnostic build path: stamps are written only when a
buffer is registered).  For each "gemm_sched" variant: sustained launches for ~1 s, then one stamped launch.
`python tools/gemm_clock.py [M N K]`"""
import ctypes
import sys
import time

import torch

sys.path.insert(0, ".")
from fastmax_experiments_amd import _lib, lora  # noqa: E402


def main():
    M, N, K = (16384, 4096, 4096) if len(sys.argv) < 4 else tuple(int(a) for a in sys.argv[1:4])
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    L = _lib.lib()
    L.fastmax_hip_debug_gemm_stamps.argtypes = [ctypes.c_void_p]
    L.fastmax_hip_debug_gemm_stamps.restype = None
    nwg = ((M + 255) // 256) * ((N + 255) // 256)
    stamps = torch.zeros(nwg, 4, dtype=torch.int64, device="cuda")
    for sched, xcd in ((0, 1), (16, 1), (17, 1), (18, 1)):
        L.fastmax_hip_tune(b"gemm_sched", sched)
        L.fastmax_hip_tune(b"gemm_xcd", xcd)
        t_end = time.perf_counter() + 1.0
        n = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while time.perf_counter() < t_end:
            for _ in range(20):
                lora.hip_gemm(x, w, None, None, None, None, N)
            n += 20
            torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / n
        L.fastmax_hip_debug_gemm_stamps(stamps.data_ptr())
        lora.hip_gemm(x, w, None, None, None, None, N)
        torch.cuda.synchronize()
        L.fastmax_hip_debug_gemm_stamps(None)
        cyc = stamps[:, 0].double().median().item()
        ticks = stamps[:, 1].double().median().item()
        lo2, lo3 = (stamps[:, 2] & 0xffffffff).double(), (stamps[:, 3] & 0xffffffff).double()
        hi2, hi3 = (stamps[:, 2] >> 32).double(), (stamps[:, 3] >> 32).double()
        wd, wb = lo2.median().item() / (K // 64), lo3.median().item() / (K // 64)
        if sched == 20:
            print(f"   4-wave kernel per K step: half 0 {hi2.median().item() / (K // 64):.0f}, wait before barrier {wd:.0f}, barrier {wb:.0f}, "
                  f"half 1 {hi3.median().item() / (K // 64):.0f}")
        print(f"sched {sched} xcd {xcd}: {ms:.3f} ms/launch = {2.0 * M * N * K / ms / 1e9:.0f} TF/s; main loop {cyc:.0f} cycles = {cyc / (K // 64):.0f} per K step "
              f"(matrix pipe: 2048), of which wave 0 waits {wd:.0f} for its copies + {wb:.0f} at the barrier; in-kernel clock {cyc / ticks * 0.1:.2f} GHz", flush=True)


if __name__ == "__main__":
    main()
