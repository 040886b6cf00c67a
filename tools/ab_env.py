#!/usr/bin/env python3
"""A/B an environment-variable kernel switch: time a few operator cases in one child process per value.
usage: python tools/ab_env.py VAR v1,v2,... [case ...]   case = op:B:H:N:D:dtype:p:mode  (mode fwd | fwd+bwd)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = ["fastmax:16:32:4096:64:bf16:2:fwd", "fastmax:16:32:4096:64:f32:2:fwd", "fastmax:8:32:2048:64:bf16:2:fwd",
           "fastmax:2:32:4096:128:bf16:2:fwd"]


def child(cases):
    sys.path.insert(0, ROOT)
    import statistics
    import torch
    from attention_mechanisms.fastmax import fastmax
    from attention_mechanisms.fastmax_hack import fastmax_hack
    for c in cases:
        op, B, H, N, D, dt, p, mode = c.split(":")
        B, H, N, D, p = int(B), int(H), int(N), int(D), int(p)
        tdt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[dt]
        g = torch.Generator(device="cuda").manual_seed(0)
        q, k, v = (torch.randn(B, H, N, D, device="cuda", generator=g).to(tdt).requires_grad_(mode != "fwd") for _ in range(3))
        f = fastmax_hack if op == "linearmax" else fastmax

        def run():
            if mode == "fwd":
                with torch.no_grad():
                    f(q, k, v, p=p, mask=True)
            else:
                o = f(q, k, v, p=p, mask=True)
                o.backward(torch.ones_like(o))
                q.grad = k.grad = v.grad = None
        for _ in range(3):
            run()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        print(f"  {c:45s} {statistics.median(ts):8.3f} ms (min {min(ts):.3f})", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2:])
    else:
        var, vals = sys.argv[1], sys.argv[2].split(",")
        cases = sys.argv[3:] or DEFAULT
        for rep in range(2):
            for v in vals:
                print(f"{var}={v} (pass {rep})", flush=True)
                env = dict(os.environ)
                env[var] = v
                subprocess.run([sys.executable, os.path.abspath(__file__), "--child", *cases], env=env, check=False)
