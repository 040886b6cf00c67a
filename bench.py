#!/usr/bin/env python3
"""Headline benchmark: fastmax attention forward (p=1, masked) at N=4096, d=64 on MI355X.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic (B,H,N,D) = (16,32,4096,64) float32
Q/K/V per GPU, already resident in HBM.  The batch dimension shards over ranks (every (b,h) pair is
independent): there is NO data-path collective; ranks only meet at the barriers that
bracket the timed region ("scaling": "weak" -- per-GPU work is fixed).  Rank 0 prints ONE JSON line.

roofline.achieved = algorithmic bytes per launch / mean launch duration, where algorithmic bytes =
4*B*H*N*D*sizeof(dtype) (read Q,K,V once, write O once) and the duration is measured
with HIP events on the stream the kernel is launched on (torch's current stream).
cpu_baseline (rank 0, N=1 only) times the CPU oracle on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s is the measured copy rate


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch B")
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--seq", type=int, default=4096)
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "f16"])
    ap.add_argument("--p", type=int, default=1)
    ap.add_argument("--mode", default="fwd", choices=["fwd", "fwd+bwd"])
    ap.add_argument("--op", default="fastmax", choices=["fastmax", "linearmax"])
    ap.add_argument("--path", default="auto", choices=["auto", "quadratic", "recurrent", "mfma"])
    ap.add_argument("--precondition-ms", type=float, default=150.0,
                    help="run the step back to back for this long before the W warm-up steps (0 = measure from an idle device)")
    ap.add_argument("--workload", default="fwd", choices=["fwd", "dp_step"],
                    help="fwd: the headline operator benchmark (default, the driver's contract); dp_step: one data-parallel "
                         "QLoRA fine-tune optimizer step per bench step (accumulation, ONE all-reduce of the LoRA-gradient bucket)")
    ap.add_argument("--config", default="tiny-llama-1.1b", choices=["pythia-14m", "tiny-llama-1.1b", "Llama-2-7b-hf"])
    ap.add_argument("--layers", type=int, default=4, help="dp_step: attention sub-layers in the stack")
    ap.add_argument("--attn", default="fastmax", choices=["fastmax", "linearmax"])
    ap.add_argument("--micro-batch", type=int, default=2)
    ap.add_argument("--accum", type=int, default=2, help="dp_step: gradient accumulation iterations per optimizer step and rank")
    ap.add_argument("--lora-dropout", type=float, default=0.05, help="dp_step: LoRA dropout (finetune/lora.py:42 default 0.05)")
    ap.add_argument("--toy", action="store_true", help="dp_step: CPU stand-in model (rehearsal of the multi-rank control flow)")
    ap.add_argument("--graph", action="store_true", help="dp_step: replay each micro-batch (forward + backward) as one captured HIP graph")
    ap.add_argument("--allow-variant", action="store_true",
                    help="A/B runs: time a non-default headline kernel (tuning key mfma_variant != 200); the line is marked "
                         "\"headline\": false.  Without it bench.py refuses such a run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def cpu_baseline(args, seconds):
    """Time the CPU oracle on a bounded sample of the headline workload (reported baseline only).

    'value' is the reference-equivalent restatement (materialise the (N,D,D) outer products and
    torch.cumsum them, as attention_mechanisms/fastmax.py:236-248 does) with all host threads on
    (1,4,N,D); the streaming plain-C port is reported beside it."""
    import numpy as np
    import torch
    from oracle import c_oracle, fastmax_oracle as orc
    N, D = args.seq, args.dim
    nt = 8.0 * (D ** 0.5)
    threads = torch.get_num_threads()
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(1, 4, N, D, generator=g) for _ in range(3))
    with torch.no_grad():
        orc.fastmax_fwd_reference_equivalent(q, k, v, nt, p=args.p)        # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            orc.fastmax_fwd_reference_equivalent(q, k, v, nt, p=args.p)
            n += 1
            el = time.perf_counter() - t0
            if (el >= seconds * 0.6 and n >= 3) or el > 3 * seconds:
                break
    ref_equiv = n * 1 * N / el
    # streaming C port, OpenMP over heads, 32 heads so every core has work
    qn, kn, vn = (np.random.default_rng(1).standard_normal((1, 32, N, D)).astype(np.float32) for _ in range(3))
    c_oracle.fwd(qn, kn, vn, p=args.p)
    n2, t0 = 0, time.perf_counter()
    while True:
        c_oracle.fwd(qn, kn, vn, p=args.p)
        n2 += 1
        el2 = time.perf_counter() - t0
        if (el2 >= seconds * 0.4 and n2 >= 2) or el2 > 3 * seconds:
            break
    return {
        "value": round(ref_equiv, 1), "unit": "tokens/s", "cores": threads, "kind": "port",
        "sample": f"oracle.fastmax_fwd_reference_equivalent (outer-product + cumsum, torch CPU ops, fp32) on "
                  f"(B,H,N,D)=(1,4,{N},{D}) p={args.p} masked, {n} runs in {el:.1f}s; tokens = B*N per run",
        "head_tokens_per_s": round(ref_equiv * 4, 1),
        "streaming_c_port": {"value": round(n2 * N / el2, 1), "unit": "tokens/s at H=32",
                             "head_tokens_per_s": round(n2 * 32 * N / el2, 1),
                             "cores": c_oracle.max_threads(),
                             "sample": f"oracle/fastmax_oracle.c (carried-state recurrence, fp64 accumulate, OpenMP over "
                                       f"heads) on (1,32,{N},{D}), {n2} runs in {el2:.1f}s"},
        "host_cpu_count": os.cpu_count(),
    }


def _baseline_metric():
    """the metric string of BASELINE.json (kept verbatim so the line can be matched against it)"""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "fastmax attn tokens/sec + achieved HBM GB/s at N=4096 d=64, 1/8 GPU"


METRIC = _baseline_metric()


def dp_step_main(args, dev, rank, world, multi, json_fd):
    """`--workload dp_step`: a bench step = one optimizer step of the data-parallel QLoRA fine-tune structure
    (finetune/lora.py:207-226): --accum micro-batches of (--micro-batch, --seq) tokens through --layers QLoRA attention
    sub-layers + lm-head loss per rank, then ONE all-reduce of the flat LoRA-gradient bucket (RCCL over xGMI) and AdamW."""
    import torch.distributed as dist
    from fastmax_experiments_amd import finetune_step
    seq = args.seq if args.seq != 4096 or args.config == "Llama-2-7b-hf" else 2048       # config 3 default: seq 2048
    res = finetune_step.run(args.config, args.layers, args.attn, seq, args.micro_batch, args.accum, args.steps, args.warmup,
                            dev, rank=rank, world=world, toy=args.toy, precondition_ms=0.0 if args.toy else args.precondition_ms,
                            graph=args.graph, lora_dropout=args.lora_dropout)
    if rank == 0:
        line = {
            "metric": "data-parallel QLoRA fine-tune step, tokens/sec (whole job)", "workload": "dp_step",
            "value": round(res["tokens_per_step"] * args.steps / res["elapsed_s"], 1), "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(res["step_ms"], 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16 (NF4 base weights)",
            "data": "synthetic",
            "config": {"workload": f"dp_step: {args.layers} x QLoRA attention sub-layer ({'toy' if args.toy else args.config}, "
                                   f"{args.attn}), seq {seq}, micro-batch {args.micro_batch} x accum {args.accum} per rank, "
                                   f"LoRA r=8 alpha=16 dropout={args.lora_dropout}, lm-head cross entropy, AdamW on the LoRA parameters",
                       "global_batch": args.micro_batch * args.accum * world,
                       "parallelism": f"dp{world} (batch sharded; one all-reduce of the flat LoRA-gradient bucket per step)"},
            "allreduce": {"ms": round(res["allreduce_ms"], 4), "bucket_bytes": res["bucket_bytes"],
                          "per_step": 1, "backend": dist.get_backend() if multi else "none (one rank)"},
            "trainable_params": res["trainable_params"], "last_loss": round(res["last_loss"], 5),
            "precondition_ms": 0.0 if args.toy else args.precondition_ms, "hip_graph": bool(args.graph and not args.toy),
        }
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if multi:
        dist.destroy_process_group()


def spawn_ranks(args):
    """`--gpus N` (N > 1) without a torch.distributed.run environment: start the N ranks the way the driver does, as child
    processes of THIS process -- before anything here has touched the GPU -- and exit with their status.  Rank 0 of the children
    prints the one JSON line on the inherited stdout."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: --gpus %d without WORLD_SIZE: launching %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    # stdout carries exactly ONE JSON line (rank 0).  Libraries write there too -- the image exports NCCL_DEBUG=VERSION and RCCL
    # prints a five-line banner (and its warnings) on stdout, NCCL_DEBUG_FILE notwithstanding -- so file descriptor 1 points
    # at stderr for the whole run and the JSON line goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the line would report the wrong number of GPUs")
    cpu_toy = args.workload == "dp_step" and args.toy
    if not torch.cuda.is_available() and not cpu_toy:
        raise SystemExit("bench.py needs an MI355X: the fastmax operator has no CPU fallback")
    # FASTMAX_BENCH_BACKEND=gloo: rehearsal of the multi-rank control flow on a box with fewer GPUs than ranks (ranks share
    # devices, barriers / MAX over gloo on the host); the measured runs use RCCL with one GPU per rank
    backend = os.environ.get("FASTMAX_BENCH_BACKEND", "nccl")
    if cpu_toy:
        backend, dev = "gloo", torch.device("cpu")
    else:
        if backend == "gloo":
            local_rank %= torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    multi = world > 1 or os.environ.get("FASTMAX_BENCH_FORCE_DIST") == "1"      # FORCE_DIST: exercise RCCL init with one rank
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)      # RCCL on ROCm; used for the barriers only

    if args.workload == "dp_step":
        return dp_step_main(args, dev, rank, world, multi, json_fd)

    from attention_mechanisms.fastmax import fastmax
    from attention_mechanisms.fastmax_hack import fastmax_hack
    from fastmax_experiments_amd import _lib, ops
    L = _lib.lib()
    ops.set_forced_path({"auto": 0, "quadratic": 1, "recurrent": 2, "mfma": 3}[args.path])
    tune_state = {k: L.fastmax_hip_tune_get(k.encode()) for k in ("mfma_variant", "bf16_kernel", "gemm_sched", "gemm_group_m", "gemm_xcd")}
    tune_state["build_flags"] = L.fastmax_hip_build_flags()           # bit 0: built with the timing-only (wrong-result) ablations
    tune_state["env"] = {k: v for k, v in sorted(os.environ.items()) if k.startswith("FASTMAX_")}
    default_kernel = tune_state["mfma_variant"] == 200 and tune_state["build_flags"] == 0
    if not default_kernel and not args.allow_variant:
        raise SystemExit(f"bench.py: the headline kernel is not the default one (mfma_variant={tune_state['mfma_variant']}, "
                         f"build_flags={tune_state['build_flags']}); pass --allow-variant for an A/B run")

    B, H, N, D = args.batch, args.heads, args.seq, args.dim
    tdt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[args.dtype]
    gen = torch.Generator(device=dev).manual_seed(rank)
    q, k, v = (torch.randn(B, H, N, D, device=dev, generator=gen).to(tdt) for _ in range(3))
    train = args.mode == "fwd+bwd"
    if train:
        go = torch.randn(B, H, N, D, device=dev, generator=gen).to(tdt)
        q.requires_grad_(True), k.requires_grad_(True), v.requires_grad_(True)

    def step():
        if args.op == "linearmax":
            o = fastmax_hack(q, k, v, p=args.p, mask=True)
        else:
            o = fastmax(q, k, v, mask=True, p=args.p)
        if train:
            q.grad = k.grad = v.grad = None
            o.backward(go)
        return o

    path = _lib.PATH_NAMES.get(ops.selected_path(q, k, args.p, True), "?")
    launches = {"n": 0}                                   # steps issued so far (each is one launch of the dominant kernel)

    def protocol():
        """W untimed warm-up steps, then EXACTLY K timed steps between barrier + synchronize pairs.
        -> (elapsed seconds, MAX over ranks; mean device-side step duration in ms from HIP events on the launch stream)"""
        for _ in range(args.warmup):
            step()
        launches["n"] += args.warmup
        launches["before_timed"] = launches["n"]
        torch.cuda.synchronize(dev)
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        t0 = time.perf_counter()
        ev[0].record()
        for i in range(args.steps):
            step()
            ev[i + 1].record()
        torch.cuda.synchronize(dev)
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        if multi:
            t = torch.tensor([el], device=dev if backend != "gloo" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        launches["n"] += args.steps
        per = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]
        return el, sum(per) / args.steps, per

    # From an idle device the chip runs the first launches at boost clock, then its power controller pulls the clock down and
    # lets it recover over ~30 ms (profiles/r02_transient.md: the same memory traffic without the matrix instructions shows
    # no such dip).  With W = 5, K = 20 the whole timed region sits inside that start-up transient, so it is measured and
    # reported on its own ("cold_start"), and the headline numbers are the SAME protocol run after --precondition-ms of
    # back-to-back steps, i.e. at the clock the chip sustains -- what a training job sees.  --precondition-ms 0 makes the
    # headline the from-idle measurement.
    cold = None
    if args.precondition_ms > 0:
        cold = protocol()
        t_end = time.perf_counter() + args.precondition_ms * 1e-3
        while time.perf_counter() < t_end:
            for _ in range(16):
                step()
            launches["n"] += 16
            torch.cuda.synchronize(dev)
    elapsed, kernel_ms, per_launch = protocol()

    if rank == 0:
        es = {"f32": 4, "bf16": 2, "f16": 2}[args.dtype]
        alg_bytes = (4 if not train else 4 + 8) * B * H * N * D * es     # fwd: Q,K,V in + O out; bwd adds Q,K,V,O,G in + dQ,dK,dV out
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tf) and not train and args.dtype == "f32" and args.p == 1 and args.op == "fastmax" \
                and (B, H, N, D) == (16, 32, 4096, 64):
            try:
                traffic = json.load(open(tf)).get(path, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        traffic_source = "profiles/hbm_traffic.json (rocprofv3 PMC passes of this command, committed; not re-measured in this run)" \
            if traffic is not None else None
        line = {
            "metric": METRIC,
            "value": round(world * B * N * args.steps / elapsed, 1),
            "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed * 1e3 / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            # every launch of the step before the first timed one: the W warm-up steps of the timed pass and, ahead of them, the
            # from-idle pass ("cold_start": W + K launches) and the --precondition-ms of back-to-back steps
            "effective_warmup_launches": launches["before_timed"],
            "headline": bool(default_kernel),
            "config": {"workload": f"{args.op} p={args.p} masked {args.mode}, (B,H,N,D)=({B},{H},{N},{D}) per GPU, "
                                   f"q,k,v~N(0,1) seed=rank, {args.dtype} I/O, fp32 accumulate",
                       "B_per_gpu": B, "H": H, "N": N, "D": D, "global_batch": B * world,
                       "parallelism": f"dp{world} (batch sharded, no data-path collective)", "kernel_path": path},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": round(kernel_ms, 4),
                         "per_launch_ms": {"min": round(min(per_launch), 4), "median": round(sorted(per_launch)[len(per_launch) // 2], 4),
                                           "max": round(max(per_launch), 4), "n": len(per_launch)},
                         "head_tokens_per_s": round(B * H * N / (kernel_ms * 1e-3), 1)},
        }
        line["precondition_ms"] = args.precondition_ms
        line["tune_state"] = tune_state
        if cold is not None:
            c_el, c_ms, c_per = cold
            line["cold_start"] = {
                "note": "the same W warm-up + K timed steps started from an idle device, before the preconditioning: the chip's "
                        "power controller dips the clock for ~30 ms after a start from idle (profiles/r02_transient.md)",
                "value": round(world * B * N * args.steps / c_el, 1), "ms_per_step": round(c_el * 1e3 / args.steps, 4),
                "effective_warmup_launches": args.warmup,
                "kernel_ms": round(c_ms, 4), "per_launch_ms": {"min": round(min(c_per), 4), "max": round(max(c_per), 4)},
                "achieved": round(alg_bytes / (c_ms * 1e-3) / 1e9, 1),
                "frac": round(alg_bytes / (c_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
