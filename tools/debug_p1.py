#!/usr/bin/env python3
"""Error anatomy of the headline forward against the C oracle (test infrastructure): per 16-query tile and per
16-column tile, for a few sequence lengths.  usage: python tools/debug_p1.py [variant ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from attention_mechanisms.fastmax import fastmax
from fastmax_experiments_amd import _lib
from oracle import c_oracle


def main():
    variants = [int(x) for x in sys.argv[1:]] or [121, 200]
    np.set_printoptions(linewidth=200, precision=2, suppress=False)
    for N in (64, 128, 256):
        g = torch.Generator().manual_seed(N)
        q, k, v = (torch.randn(1, 1, N, 64, generator=g) for _ in range(3))
        ro, rg = c_oracle.fwd(q.numpy(), k.numpy(), v.numpy())
        for var in variants:
            _lib.check(_lib.lib().fastmax_hip_tune(b"mfma_variant", var), "tune")
            o = fastmax(q.cuda(), k.cuda(), v.cuda()).cpu().numpy()
            err = np.abs(o - ro)[0, 0]                                  # (N, 64)
            tiles = err.reshape(N // 16, 16, 4, 16).max(axis=(1, 3))    # [query tile][d tile]
            print(f"N={N} variant={var}: max err {err.max():.3e} (ref max {np.abs(ro).max():.2f})")
            print(tiles)
            if err.max() > 1e-3:
                i, d = np.unravel_index(err.argmax(), err.shape)
                print(f"  worst at query {i}, d {d}: got {o[0, 0, i, d]:.5f} want {ro[0, 0, i, d]:.5f}")
                print("  row 0 got ", o[0, 0, 0, :8], "\n  row 0 want", ro[0, 0, 0, :8])
                print("  row 17 got ", o[0, 0, 17, :8], "\n  row 17 want", ro[0, 0, 17, :8])


if __name__ == "__main__":
    main()
