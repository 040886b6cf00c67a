"""N>1 path on CPU: world_size-2 gloo processes run the DataParallelStepper (flat LoRA-grad bucket, one
all-reduce per optimizer step, gradient accumulation) and must land on exactly the parameters a single
process gets from the same global batch.  The model here is a tiny stand-in with `lora_` parameters: the
attention operator itself has no CPU path and needs no collective."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class TinyLoRA(torch.nn.Module):
    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        self.base = torch.nn.Parameter(torch.randn(16, 16, generator=g))
        self.lora_A = torch.nn.Parameter(torch.randn(4, 16, generator=g) * 0.1)
        self.lora_B = torch.nn.Parameter(torch.randn(16, 4, generator=g) * 0.1)

    def forward(self, x):
        return x @ self.base.T + (x @ self.lora_A.T) @ self.lora_B.T


def _loss(model, batch):
    x, y = batch
    return ((model(x) - y) ** 2).mean()


def _data():
    g = torch.Generator().manual_seed(1)
    return torch.randn(3, 16, 16, generator=g), torch.randn(3, 16, 16, generator=g)   # 3 optimizer steps x 16 samples


def _run(rank, world, port, out):
    from fastmax_experiments_amd import dp
    if world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    model = TinyLoRA()
    params = dp.trainable_lora_parameters(model)
    assert len(params) == 2 and not model.base.requires_grad
    opt = torch.optim.AdamW(params, lr=1e-2)
    train = dp.TrainArgs(global_batch_size=16, micro_batch_size=2)
    st = dp.DataParallelStepper(model, opt, train, _loss)
    assert st.accum == 16 // world // 2
    X, Y = _data()
    for s in range(X.shape[0]):
        xs, ys = dp.shard_batch(X[s], rank, world), dp.shard_batch(Y[s], rank, world)
        for m in range(st.accum):
            st.micro_step((xs[2 * m:2 * m + 2], ys[2 * m:2 * m + 2]))
    assert st.step_count == 3
    if rank == 0:
        torch.save({k: v.detach().clone() for k, v in model.state_dict().items()}, out)
    if world > 1:
        # every rank holds identical parameters after synchronised steps
        flat = torch.cat([p.detach().flatten() for p in params])
        ref = flat.clone()
        dist.broadcast(ref, 0)
        assert torch.equal(flat, ref)
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_dp_world2_matches_single_process(tmp_path):
    single, multi = str(tmp_path / "single.pt"), str(tmp_path / "multi.pt")
    _run(0, 1, 0, single)
    mp.spawn(_run, args=(2, _free_port(), multi), nprocs=2, join=True)
    a, b = torch.load(single), torch.load(multi)
    for k in a:
        assert torch.allclose(a[k], b[k], rtol=1e-5, atol=1e-6), k
    assert not torch.equal(a["lora_A"], TinyLoRA().lora_A)          # it did train
    assert torch.equal(a["base"], TinyLoRA().base)                   # frozen base untouched


def test_flat_bucket_aliases_grads():
    from fastmax_experiments_amd import dp
    m = TinyLoRA()
    params = dp.trainable_lora_parameters(m)
    b = dp.FlatGradBucket(params)
    assert b.flat.numel() == 4 * 16 + 16 * 4 and b.nbytes == 128 * 4
    _loss(m, (torch.ones(2, 16), torch.zeros(2, 16))).backward()
    assert m.lora_A.grad.data_ptr() == b.flat.data_ptr()            # autograd wrote into the bucket
    assert float(b.flat.abs().sum()) > 0
    b.zero()
    assert float(b.flat.abs().sum()) == 0 and m.lora_A.grad.data_ptr() == b.flat.data_ptr()


def test_mixed_dtype_bucket():
    from fastmax_experiments_amd import dp
    m = TinyLoRA().to(torch.bfloat16)
    params = dp.trainable_lora_parameters(m)
    b = dp.FlatGradBucket(params, dtype=torch.float32)
    _loss(m, (torch.ones(2, 16, dtype=torch.bfloat16), torch.zeros(2, 16, dtype=torch.bfloat16))).backward()
    b.all_reduce_mean()
    assert m.lora_A.grad.dtype == torch.bfloat16 and float(b.flat.abs().sum()) > 0


def test_graph_capture_refuses_parameters_outside_the_bucket_dtype():
    """capture() records AccumulateGrad nodes that write each parameter's .grad: that only stays valid when every .grad IS a
    view of the bucket (a bf16 parameter in an fp32 bucket gets a tensor of its own, which gather() drops)"""
    from fastmax_experiments_amd.dp import DataParallelStepper, TrainArgs
    m = TinyLoRA()
    for n, p in m.named_parameters():
        p.requires_grad = "lora_" in n
    m.lora_A.data = m.lora_A.data.to(torch.bfloat16)
    opt = torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=0.1)
    st = DataParallelStepper(m, opt, TrainArgs(global_batch_size=4, micro_batch_size=4), _loss, bucket_dtype=torch.float32)
    with pytest.raises(RuntimeError, match="bucket's dtype"):
        st.capture((torch.zeros(4, 16), torch.zeros(4, 16)))


def test_accumulation_iters_follow_reference_args():
    from fastmax_experiments_amd import dp
    t = dp.TrainArgs(global_batch_size=64, micro_batch_size=4)
    assert t.gradient_accumulation_iters(1) == 16 and t.gradient_accumulation_iters(8) == 2
    with pytest.raises(AssertionError):
        dp.TrainArgs(global_batch_size=8, micro_batch_size=4).gradient_accumulation_iters(8)


def _run_bench(args, nproc, env=None):
    import json
    import subprocess
    cmd = [sys.executable]
    if nproc > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port())]
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + args
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=e, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout                      # rank 0 prints exactly ONE JSON line
    return json.loads(lines[0])


def test_bench_dp_step_mode_two_ranks_over_gloo():
    """`bench.py --workload dp_step` launched the way the driver launches N > 1 (torch.distributed.run, one process per
    rank): the multi-rank control flow -- sharded batch, accumulation, one all-reduce per optimizer step, MAX over ranks,
    one JSON line -- rehearsed on CPU ranks with the stand-in model (the attention operator has no CPU path)."""
    one = _run_bench(["--workload", "dp_step", "--toy", "--steps", "3", "--warmup", "1", "--seq", "16"], 1)
    two = _run_bench(["--workload", "dp_step", "--toy", "--steps", "3", "--warmup", "1", "--seq", "16"], 2)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["workload"] == "dp_step"
    assert two["config"]["global_batch"] == 2 * one["config"]["global_batch"]            # weak scaling: per-rank work fixed
    assert two["allreduce"]["backend"] == "gloo" and two["allreduce"]["per_step"] == 1
    assert two["allreduce"]["bucket_bytes"] == one["allreduce"]["bucket_bytes"] == 4 * two["trainable_params"]
    assert two["value"] > 0 and two["ms_per_step"] > 0 and two["last_loss"] == two["last_loss"]


def test_bench_gpus_without_a_launcher_spawns_its_ranks_or_fails():
    """`python bench.py --gpus N` started WITHOUT torch.distributed.run (no WORLD_SIZE) must not quietly run one rank and report
    n_gpus 1: it starts the N ranks itself (before touching a GPU) and passes their one JSON line through; a WORLD_SIZE that
    disagrees with --gpus exits non-zero."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "dp_step", "--toy", "--steps", "2", "--warmup", "1",
           "--seq", "16"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    assert json.loads(lines[0])["n_gpus"] == 2
    bad = subprocess.run(cmd, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in bad.stderr
