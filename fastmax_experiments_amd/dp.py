"""Data-parallel step for LoRA / QLoRA fine-tuning with the fastmax operator.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm; "gloo" for the CPU
tests).  The attention operator itself needs no collective -- every (b,h) pair is independent -- so data
parallelism is: shard the batch over ranks, accumulate micro-batch gradients locally, and at the
accumulation boundary run ONE all-reduce over a single flat bucket that aliases every trainable (LoRA)
gradient, then divide by the world size.

Mirrors the step structure of the reference's `finetune/lora.py:fit` (207-226):
  gradient_accumulation_iters = global_batch_size // devices // micro_batch_size   (lit_gpt/args.py:46-57)
  `fabric.no_backward_sync(enabled=is_accumulating)`  -> no collective while accumulating
  `fabric.backward(loss / gradient_accumulation_iters)`
  optimizer.step(); optimizer.zero_grad(); scheduler.step()  at the boundary
where the reference relies on Lightning Fabric (FSDP reduce-scatter inside backward), and refuses
quantisation together with devices > 1 (finetune/lora.py:80-85).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Iterable, List, Optional

import torch
import torch.distributed as dist


def trainable_lora_parameters(module: torch.nn.Module) -> List[torch.nn.Parameter]:
    """`mark_only_lora_as_trainable` (lit_gpt/lora.py:450-452): only names containing 'lora_' train."""
    out = []
    for name, p in module.named_parameters():
        p.requires_grad = "lora_" in name
        if p.requires_grad:
            out.append(p)
    return out


class FlatGradBucket:
    """One contiguous gradient buffer; every parameter's `.grad` is a view into it.

    Llama-2-7B, r=8 on q,v: 4,194,304 parameters = 16 MiB fp32 -> a ring all-reduce over xGMI
    (7 links x ~153 GB/s, per-link bound) costs ~0.19 ms; one flat bucket per optimizer step, no overlap needed.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], dtype: Optional[torch.dtype] = None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatGradBucket needs at least one trainable parameter")
        dev = self.params[0].device
        self.dtype = dtype or torch.float32
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=self.dtype, device=dev)
        self._views = []
        off = 0
        for p in self.params:
            v = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
            self._views.append(v)
            if p.dtype == self.dtype:
                p.grad = v                       # autograd accumulates straight into the bucket
        self._aliased = [p.dtype == self.dtype for p in self.params]

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * self.flat.element_size()

    def gather(self):
        """Copy gradients of parameters whose dtype differs from the bucket's (e.g. bf16 params, fp32 bucket)."""
        for p, v, al in zip(self.params, self._views, self._aliased):
            if al:
                if p.grad is not None and p.grad.data_ptr() != v.data_ptr():   # someone replaced .grad
                    v.copy_(p.grad)
                    p.grad = v
            elif p.grad is not None:
                v.add_(p.grad.to(self.dtype))
                p.grad = None

    def scatter(self):
        for p, v, al in zip(self.params, self._views, self._aliased):
            if not al:
                p.grad = v.to(p.dtype)

    def zero(self):
        self.flat.zero_()
        for p, v, al in zip(self.params, self._views, self._aliased):
            p.grad = v if al else None

    def all_reduce_mean(self, group=None):
        """The one data-path collective of a step: sum over ranks, then divide by the world size."""
        self.gather()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.div_(dist.get_world_size(group))
        self.scatter()


@dataclass
class TrainArgs:
    """Subset of lit_gpt/args.py:TrainArgs that shapes a step."""
    global_batch_size: int = 64
    micro_batch_size: int = 4
    max_norm: Optional[float] = None

    def batch_size(self, devices: int) -> int:
        b = self.global_batch_size // devices
        assert b > 0
        return b

    def gradient_accumulation_iters(self, devices: int) -> int:
        n = self.batch_size(devices) // self.micro_batch_size
        assert n > 0
        return n


class DataParallelStepper:
    """Runs micro-batches; synchronises and steps at the accumulation boundary (finetune/lora.py:214-226)."""

    def __init__(self, model: torch.nn.Module, optimizer: torch.optim.Optimizer, train: TrainArgs,
                 loss_fn: Callable[[torch.nn.Module, object], torch.Tensor], scheduler=None, group=None,
                 bucket_dtype: Optional[torch.dtype] = None, time_comm: bool = False):
        self.model, self.optimizer, self.train, self.loss_fn = model, optimizer, train, loss_fn
        self.scheduler, self.group = scheduler, group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.accum = train.gradient_accumulation_iters(self.world)
        self.bucket = FlatGradBucket([p for p in model.parameters() if p.requires_grad], dtype=bucket_dtype)
        self.iter_num = 0
        self.step_count = 0
        # time_comm: bracket the one collective of every optimizer step (device events on a HIP device, host clock else)
        self.time_comm = time_comm
        self.comm_ms: list = []

    def _timed_all_reduce(self):
        if not self.time_comm:
            return self.bucket.all_reduce_mean(self.group)
        if self.bucket.flat.device.type == "cuda":
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.bucket.all_reduce_mean(self.group)
            e1.record()
            self.comm_ms.append((e0, e1))                 # resolved lazily: no host sync inside the step
        else:
            import time
            t0 = time.perf_counter()
            self.bucket.all_reduce_mean(self.group)
            self.comm_ms.append((time.perf_counter() - t0) * 1e3)

    def comm_times_ms(self):
        """milliseconds of every timed all-reduce so far (synchronises the device events)"""
        return [t if isinstance(t, float) else t[0].elapsed_time(t[1]) for t in self.comm_ms]

    def capture(self, sample_batch) -> None:
        """Record one micro-batch (forward + backward of loss / accum into the bucket) as a HIP graph; `micro_step` then
        copies its batch into the recorded input tensors and replays.  The step is ~600 kernel launches of 5-500 us each:
        replayed as one graph the GPU no longer waits for the host between them.  The collective and the optimizer stay
        outside the graph.  Needs static shapes (every micro-batch like `sample_batch`) and a step without host
        synchronisation -- which the HIP path is (workspaces come from the caching allocator, scalars stay on the device)."""
        dev = self.bucket.flat.device
        if not all(self.bucket._aliased):
            # a parameter whose dtype differs from the bucket's gets a separate .grad tensor that the recorded AccumulateGrad
            # would keep writing to after gather() has dropped it (and the warm-up gradients would enter the first step)
            raise RuntimeError("graph capture needs every trainable parameter in the bucket's dtype "
                               f"({self.bucket.dtype}): its .grad must be a view of the bucket")
        if dev.type != "cuda":
            raise RuntimeError("graph capture needs a HIP device")
        self._static = tuple(t.clone() for t in sample_batch)
        saved = self.bucket.flat.clone()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                      # warm-up on a side stream: lazy initialisations, allocator pools
            for _ in range(2):
                (self.loss_fn(self.model, self._static) / self.accum).backward()
        torch.cuda.current_stream(dev).wait_stream(side)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._static_loss = self.loss_fn(self.model, self._static)
            (self._static_loss / self.accum).backward()
        self.bucket.flat.copy_(saved)                      # warm-up and capture accumulated into the bucket: undo

    _graph = None

    def micro_step(self, batch) -> torch.Tensor:
        """One micro-batch: forward, backward of loss/accum; at the boundary all-reduce + optimizer step.
        Returns the (unscaled) micro-batch loss."""
        self.iter_num += 1
        is_accumulating = self.iter_num % self.accum != 0
        if self._graph is not None:
            for dst, src in zip(self._static, batch):
                dst.copy_(src)
            self._graph.replay()
            loss = self._static_loss.detach().clone()      # the recorded tensor is overwritten by the next replay
        else:
            loss = self.loss_fn(self.model, batch)
            (loss / self.accum).backward()
        if not is_accumulating:
            self._timed_all_reduce()
            if self.train.max_norm is not None:
                torch.nn.utils.clip_grad_norm_(self.bucket.params, self.train.max_norm)
            self.optimizer.step()
            self.bucket.zero()                    # optimizer.zero_grad(): keep the views aliased
            if self.scheduler is not None:
                self.scheduler.step()
            self.step_count += 1
        return loss.detach()


def shard_batch(global_batch: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Even split of the batch dimension over ranks (the only partitioning the path needs)."""
    assert global_batch.shape[0] % world == 0, "global batch must divide by the number of ranks"
    per = global_batch.shape[0] // world
    return global_batch[rank * per:(rank + 1) * per]
