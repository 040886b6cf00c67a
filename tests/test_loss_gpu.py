"""Loss path: the HIP cross-entropy kernels behind chunked_cross_entropy and the lm-head + loss
pair, against torch.nn.functional.cross_entropy in float64 / the reference's own formulation (lit_gpt/utils.py:228-272)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")


def _reference_chunked(logits, targets, chunk_size, ignore_index):
    """the reference's arithmetic (utils.py:228-272) in float64 on the same inputs"""
    if isinstance(logits, list):
        logits = torch.cat(logits, dim=1)
    l2 = logits.reshape(-1, logits.size(-1)).double()
    t = targets.reshape(-1)
    rows = F.cross_entropy(l2, t, ignore_index=ignore_index, reduction="none")
    return rows.sum() / max(1, int((t != ignore_index).sum()))


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-6), (torch.bfloat16, 6e-3), (torch.float16, 1e-3)])
@pytest.mark.parametrize("B,T,V,as_list", [(2, 300, 32000, True), (1, 129, 50257, False), (3, 64, 1000, True), (2, 17, 33, False)])
def test_chunked_cross_entropy_forward_backward(B, T, V, as_list, dt, tol):
    from fastmax_experiments_amd.loss import chunked_cross_entropy
    g = torch.Generator().manual_seed(V + T)
    logits = (torch.randn(B, T, V, generator=g) * 3).to(dt).cuda()
    targets = torch.randint(0, V, (B, T), generator=g).cuda()
    targets[0, ::7] = -1                                            # ignored positions
    ref_in = logits.double().requires_grad_(True)
    ref = _reference_chunked(ref_in, targets, 128, -1)
    ref.backward()
    x = logits.clone().requires_grad_(True)
    arg = list(x.split(128, dim=1)) if as_list else x
    loss = chunked_cross_entropy(arg, targets, chunk_size=128, ignore_index=-1)
    loss.backward()
    assert abs(float(loss) - float(ref)) <= tol * max(1.0, abs(float(ref)))
    gerr = (x.grad.double() - ref_in.grad).abs().max() / ref_in.grad.abs().max()
    assert float(gerr) < (1e-5 if dt == torch.float32 else 1e-2)
    assert x.grad.dtype == dt
    assert float(x.grad[0, 0].abs().sum()) == 0.0                   # ignored row


def test_all_targets_ignored_gives_zero():
    from fastmax_experiments_amd.loss import chunked_cross_entropy
    logits = torch.randn(1, 8, 100, device="cuda", requires_grad=True)
    targets = torch.full((1, 8), -1, device="cuda")
    loss = chunked_cross_entropy(logits, targets)
    loss.backward()
    assert float(loss) == 0.0 and float(logits.grad.abs().sum()) == 0.0


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
def test_lm_head_cross_entropy_matches_the_unfused_pair(dt, tol):
    """x -> lm_head -> chunked_cross_entropy (finetune/lora.py:216-219) with logits alive one row block at a time"""
    from fastmax_experiments_amd.loss import lm_head_cross_entropy
    g = torch.Generator().manual_seed(3)
    M, K, V = 1000, 256, 5000
    x = torch.randn(2, M // 2, K, generator=g).to(dt).cuda()
    w = (torch.randn(V, K, generator=g) * 0.05).to(dt).cuda()
    t = torch.randint(0, V, (2, M // 2), generator=g).cuda()
    t[1, :50] = -1
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    ref = _reference_chunked(xr @ wr.t(), t, 128, -1)
    ref.backward()
    xa, wa = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    loss = lm_head_cross_entropy(xa, wa, t, ignore_index=-1, chunk_rows=384)
    loss.backward()
    assert abs(float(loss) - float(ref)) <= tol * abs(float(ref))
    for got, want in ((xa.grad, xr.grad), (wa.grad, wr.grad)):
        err = (got.double() - want).abs().max() / want.abs().max()
        assert float(err) < (1e-4 if dt == torch.float32 else 3e-2)


def test_cpu_tensors_are_refused():
    from fastmax_experiments_amd.loss import chunked_cross_entropy
    with pytest.raises(RuntimeError):
        chunked_cross_entropy(torch.randn(1, 4, 10), torch.zeros(1, 4, dtype=torch.long))


def test_labels_outside_the_vocabulary():
    """-100 labels under ignore_index=-1 (a common mix-up): torch raises a device assert; here the kernel scores such rows 0
    with a zero gradient AND the mean's denominator leaves them out (it used to count them: a silently smaller loss);
    `check_targets` raises like the reference."""
    from fastmax_experiments_amd.loss import check_targets, chunked_cross_entropy, lm_head_cross_entropy
    g = torch.Generator().manual_seed(2)
    V = 96
    logits = torch.randn(2, 40, V, generator=g).cuda().requires_grad_(True)
    targets = torch.randint(0, V, (2, 40), generator=g).cuda()
    targets[0, :5] = -100
    targets[1, 3] = V + 7
    targets[1, 4] = -1                                               # the real ignore_index
    loss = chunked_cross_entropy(logits, targets, chunk_size=0, ignore_index=-1)
    keep = (targets >= 0) & (targets < V)
    ref = F.cross_entropy(logits.detach()[keep], targets[keep], reduction="mean")
    assert abs(float(loss) - float(ref)) < 1e-5
    loss.backward()
    assert float(logits.grad[0, :5].abs().sum()) == 0.0 and float(logits.grad[1, 3].abs().sum()) == 0.0
    with pytest.raises(ValueError):
        check_targets(targets, V, ignore_index=-1)
    check_targets(targets.clamp(0, V - 1), V, ignore_index=-1)
    x = torch.randn(80, 32, generator=g).cuda()
    w = torch.randn(V, 32, generator=g).cuda()
    l2 = lm_head_cross_entropy(x, w, targets.reshape(-1), ignore_index=-1, chunk_rows=32)
    ref2 = F.cross_entropy((x @ w.T)[keep.reshape(-1)], targets.reshape(-1)[keep.reshape(-1)], reduction="mean")
    assert abs(float(l2) - float(ref2)) < 2e-3


def test_kept_logits_are_consumed_once():
    """the default fine-tune route keeps the bf16 logits and turns them into d(logits) in place: same loss and gradient as
    the recompute route, and a second backward pass through the same graph is refused instead of using the overwritten buffer"""
    from fastmax_experiments_amd.loss import _LMHeadLoss
    g = torch.Generator().manual_seed(11)
    x = torch.randn(300, 128, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(1000, 128, generator=g) * 0.1).to(torch.bfloat16).cuda()
    t = torch.randint(0, 1000, (300,), generator=g).cuda()
    res = []
    for keep in (False, True):
        xa = x.clone().requires_grad_(True)
        loss = _LMHeadLoss.apply(xa, w, t, -1, 128, keep)
        loss.backward(retain_graph=True)
        res.append((float(loss), xa.grad.float().clone()))
        if keep:
            with pytest.raises(RuntimeError):
                loss.backward()
    assert res[0][0] == res[1][0]
    assert float((res[0][1] - res[1][1]).abs().max()) <= 1e-2 * float(res[0][1].abs().max())
