#!/usr/bin/env python3
"""Randomised parity hunt: fastmax forward + backward on the HIP path against the dense float64 oracle over random shapes,
dtypes, orders, masks and memory layouts (contiguous, (B,N,H,D)-transposed views, grouped-query stride-0 head views, sliced
head sizes).  Prints every case above its tolerance; exit code 1 if any.  usage: fuzz_parity.py [cases] [seed]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from attention_mechanisms.fastmax import fastmax
from attention_mechanisms.fastmax_hack import fastmax_hack
from oracle import fastmax_oracle as orc

DT = {"f32": (torch.float32, 2e-4), "bf16": (torch.bfloat16, 3e-2), "f16": (torch.float16, 4e-3)}


def nw(a, r):
    """normwise relative error; against an (almost) zero reference -- e.g. dq, dk of a single-token sequence -- the absolute one"""
    nr = float(np.linalg.norm(r))
    return float(np.linalg.norm(a - r)) / (nr if nr > 1e-6 * np.sqrt(r.size) else 1.0)


def layout(t, kind, rng):
    """the same values behind a different memory layout"""
    if kind == "contig":
        return t.contiguous()
    if kind == "bnhd":                                   # (B,N,H,D) storage viewed as (B,H,N,D): the qkv-split's natural view
        return t.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)
    if kind == "padded":                                 # rows of a wider tensor (row stride > D, 16-byte aligned)
        B, H, N, D = t.shape
        big = torch.zeros(B, H, N, D + 8, dtype=t.dtype, device=t.device)
        big[..., :D] = t
        return big[..., :D]
    raise ValueError(kind)


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = random.Random(seed)
    bad = 0
    for ci in range(ncases):
        dt = rng.choice(["f32", "bf16", "f16", "f32", "bf16"])
        tdt, tol = DT[dt]
        p = rng.choice([1, 2])
        mask = rng.random() < 0.7
        D = rng.choice([8, 16, 24, 32, 40, 48, 64, 64, 64, 80, 96, 128, 128, 160, 192, 256])
        B, H = rng.choice([(1, 1), (1, 2), (2, 3), (1, 4), (3, 2)])
        Nq = rng.choice([1, 7, 16, 63, 64, 65, 127, 200, 256, 257, 511, 512, 513, 700, 1000, 1024, 1536, 2047])
        Nk = Nq if mask else rng.choice([Nq, Nq, max(1, Nq // 2), Nq + 37, 300])
        rep = rng.choice([1, 1, 2, 4]) if H % 2 == 0 or H == 1 else 1
        kinds = [rng.choice(["contig", "contig", "bnhd", "padded"]) for _ in range(3)]
        g = torch.Generator().manual_seed(seed * 100003 + ci)
        q = torch.randn(B, H, Nq, D, generator=g).to(tdt)
        gk = H // rep if (H % rep == 0 and rep > 1) else H
        k = torch.randn(B, gk, Nk, D, generator=g).to(tdt)
        v = torch.randn(B, gk, Nk, D, generator=g).to(tdt)
        go = torch.randn(B, H, Nq, D, generator=g).to(tdt)
        qd = layout(q.cuda(), kinds[0], rng).requires_grad_(True)
        kd0 = layout(k.cuda(), kinds[1], rng).requires_grad_(True)
        vd0 = layout(v.cuda(), kinds[2], rng).requires_grad_(True)
        views = False
        if gk != H:                                       # grouped-query: every query head of a group reads the same K, V rows
            rep_ = H // gk
            kf = k.repeat_interleave(rep_, dim=1)
            vf = v.repeat_interleave(rep_, dim=1)
            views = kinds == ["contig"] * 3 and rng.random() < 0.7
            if views:                                     # (batch x group) as the batch axis, K / V heads with stride 0: nothing copied
                qd_in = qd.view(B * gk, rep_, Nq, D)
                kd = kd0.view(B * gk, 1, Nk, D).expand(B * gk, rep_, Nk, D)
                vd = vd0.view(B * gk, 1, Nk, D).expand(B * gk, rep_, Nk, D)
            else:
                qd_in, kd, vd = qd, kd0.repeat_interleave(rep_, dim=1), vd0.repeat_interleave(rep_, dim=1)
        else:
            qd_in, kd, vd, kf, vf = qd, kd0, vd0, k, v
        if gk == H and mask and D <= 128 and Nq >= 2 and rng.random() < 0.25:
            # linearmax WITH gradients (masked, p = 1: at N >= 512 the one-node route with the prologue and its backward inside
            # the scans) against float64 autograd over the definition (fastmax_hack.py:36-60 as dense masked first-order attention)
            desc = f"case {ci}: linearmax train {dt} (B,H,N,D)=({B},{H},{Nq},{D}) layouts={kinds}"
            try:
                o = fastmax_hack(qd, kd, vd, p=1, mask=True)
                o.backward(go.cuda().to(o.dtype))
            except Exception as e:                        # noqa: BLE001
                print("RAISED", desc, type(e).__name__, str(e)[:200], flush=True)
                bad += 1
                continue
            q64, k64, v64 = (t.double().requires_grad_(True) for t in (q, k, v))
            def nrm(x):
                xc = x - x.mean(-1, keepdim=True)
                return xc / xc.norm(dim=-1).amax(-1)[..., None, None]
            sc = torch.tril(1.0 + nrm(q64) @ nrm(k64).transpose(-1, -2))
            ref = (sc @ v64) / sc.sum(-1, keepdim=True)
            ref.backward(go.double())
            errs = {"o": nw(o.detach().double().cpu().numpy(), ref.detach().numpy()), "dq": nw(qd.grad.double().cpu().numpy(), q64.grad.numpy()),
                    "dk": nw(kd0.grad.double().cpu().numpy(), k64.grad.numpy()), "dv": nw(vd0.grad.double().cpu().numpy(), v64.grad.numpy())}
            ok = max(errs.values()) <= tol
            bad += 0 if ok else 1
            print("ok  " if ok else "BAD ", desc, " ".join(f"{n}={e:.2e}" for n, e in errs.items()), flush=True)
            continue
        if gk == H and D <= 128 and rng.random() < 0.2 and 2 <= Nq <= 700 and Nk <= 700:      # (the numpy restatement walks 32-token chunks)
            # linearmax (fastmax_hack.py:5-60), forward: prologue + operator against the oracle's restatement
            desc = f"case {ci}: linearmax {dt} p={p} mask={mask} (B,H,Nq,Nk,D)=({B},{H},{Nq},{Nk},{D}) layouts={kinds}"
            try:
                with torch.no_grad():
                    o = fastmax_hack(qd, kd, vd, p=p, mask=mask)
            except Exception as e:                        # noqa: BLE001
                print("RAISED", desc, type(e).__name__, str(e)[:200], flush=True)
                bad += 1
                continue
            ro = np.asarray(orc.linearmax_fwd(q.double().numpy(), k.double().numpy(), v.double().numpy(), p=p, mask=mask))
            e_o = nw(o.double().cpu().numpy(), ro)
            ok = e_o <= tol
            bad += 0 if ok else 1
            print("ok  " if ok else "BAD ", desc, f"o={e_o:.2e}", flush=True)
            continue
        desc = f"case {ci}: {dt} p={p} mask={mask} (B,H,Nq,Nk,D)=({B},{H},{Nq},{Nk},{D}) groups={gk} views={views} layouts={kinds}"
        try:
            o = fastmax(qd_in, kd, vd, mask=mask, p=p).reshape(B, H, Nq, D)
            o.backward(go.cuda().to(o.dtype))
        except Exception as e:                            # noqa: BLE001
            print("RAISED", desc, type(e).__name__, str(e)[:200], flush=True)
            bad += 1
            continue
        qn, kn, vn, gn = (t.double().numpy() for t in (q, kf, vf, go))
        ro, _ = orc.fastmax_fwd_dense(qn, kn, vn, mask=mask, p=p)
        dq, dk, dv = orc.fastmax_bwd_dense(qn, kn, vn, gn, mask=mask, p=p)
        if gk != H:                                       # gradients of the shared rows: sum over the group
            dk = dk.reshape(B, gk, H // gk, Nk, D).sum(2)
            dv = dv.reshape(B, gk, H // gk, Nk, D).sum(2)
        errs = {"o": nw(o.detach().double().cpu().numpy(), ro), "dq": nw(qd.grad.double().cpu().numpy(), dq),
                "dk": nw(kd0.grad.double().cpu().numpy(), dk), "dv": nw(vd0.grad.double().cpu().numpy(), dv)}
        # 16-bit gradients of shared rows are sums of 16-bit-rounded terms: allow for that
        worst = max(errs.values())
        flag = "BAD " if not (worst <= tol) else "ok  "
        if flag == "BAD ":
            bad += 1
        print(flag, desc, " ".join(f"{k_}={v_:.2e}" for k_, v_ in errs.items()), flush=True)
    print(f"{ncases - bad} / {ncases} within tolerance", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
