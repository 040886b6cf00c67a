#!/usr/bin/env python3
"""Where a wave of the p=2 forward kernel (fastmax_quad32_mfma.hip, D <= 64) spends its time: issue-time split of the tile
loop from s_memtime stamps.  Needs the diagnostic build:
    FASTMAX_HIPCC_EXTRA=-DFASTMAX_QUAD32_STAMPS python -c "from fastmax_experiments_amd import build; build.build(force=True)"
usage: quad32_stamps.py [B H N D]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastmax_experiments_amd import _lib, ops


def main():
    B, H, N, D = (16, 32, 4096, 64) if len(sys.argv) < 5 else tuple(int(a) for a in sys.argv[1:5])
    L = _lib.lib()
    L.fastmax_hip_debug_quad32_stamps.argtypes = [ctypes.c_void_p]
    L.fastmax_hip_debug_quad32_stamps.restype = ctypes.c_int
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v = (torch.randn(B, H, N, D, device="cuda", generator=g).to(torch.bfloat16) for _ in range(3))
    nwg = B * H * ((N + 127) // 128)
    stamps = torch.zeros(nwg, 8, dtype=torch.int64, device="cuda")
    for _ in range(20):
        ops.forward(q, k, v, 2, True, 1.0, 0.0, torch.bfloat16)
    torch.cuda.synchronize()
    assert L.fastmax_hip_debug_quad32_stamps(stamps.data_ptr()) == 0
    ops.forward(q, k, v, 2, True, 1.0, 0.0, torch.bfloat16)
    torch.cuda.synchronize()
    L.fastmax_hip_debug_quad32_stamps(None)
    s = stamps.double()
    s = s[s[:, 0] > 8]                                   # workgroups with a fair number of unmasked tiles
    tiles = s[:, 0]
    names = ["QK issue (8 MFMA + K fragment reads)", "V^T reads + polynomial + pack (2 x)", "PV issue (2 x 4 MFMA)",
             "request tile t+2 (address arithmetic + 4 global loads)", "barrier (+ tail of the PV MFMAs)",
             "wait for tile t+1's global loads", "commit tile t+1 (4 ds_write_b128)"]
    tot = 0.0
    print(f"(B,H,N,D)=({B},{H},{N},{D}) bf16 p=2: {int(s.shape[0])} workgroups, wave 0, cycles per unmasked 64-key tile (medians)")
    for i, nm in enumerate(names):
        val = (s[:, 1 + i] / tiles).median().item()
        tot += val
        print(f"  {nm:45s} {val:8.0f}")
    print(f"  {'sum':45s} {tot:8.0f}   (matrix pipe needs 16 x 32 = 512 per wave-tile; three waves share a SIMD)")


if __name__ == "__main__":
    main()
