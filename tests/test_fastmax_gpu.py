"""Parity tests proper: the HIP path (through the reference's Python signature -> ctypes -> C ABI)
against the golden vectors from the reference and against the CPU oracle.  Need an MI355X."""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden, rel_err

pytestmark = pytest.mark.gpu

# float32 kernels accumulate in fp32 (the matrix-core path in split-bf16, ~16 mantissa bits per
# product); the north star asks for 1e-3 relative -- the tests hold the fp32 paths to 2e-4 / 1e-3
TOL_FWD = 2e-4
TOL_BWD = 1e-3


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from fastmax_experiments_amd import _lib
    _lib.lib()          # fail loudly if the extension is missing


def _t(x, dtype=torch.float32, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(x)).to("cuda", dtype)
    return t.requires_grad_(grad)


def _kw(meta):
    return dict(meta.get("kw", {}))


PATHS = ["auto", "quadratic", "recurrent", "mfma", "quadratic_mfma"]


def _force(path):
    from fastmax_experiments_amd import _lib, ops
    ops.set_forced_path({"auto": _lib.PATH_AUTO, "quadratic": _lib.PATH_QUADRATIC,
                         "recurrent": _lib.PATH_RECURRENT, "mfma": _lib.PATH_MFMA,
                         "quadratic_mfma": _lib.PATH_QUADRATIC_MFMA}[path])


@pytest.fixture(autouse=True)
def _restore_path():
    yield
    _force("auto")


@pytest.mark.parametrize("name", golden_names("fm_") + golden_names("opt_"))
def test_golden_forward_backward(name):
    from attention_mechanisms.fastmax import fastmax
    d, meta = load_golden(name)
    q, k, v = (_t(d[n], grad=True) for n in "qkv")
    o = fastmax(q, k, v, mask=meta["mask"], p=meta["p"], **_kw(meta))
    assert o.dtype == torch.float32 and o.shape == q.shape and o.is_contiguous()
    assert rel_err(o.detach().cpu().numpy(), d["o"]) < TOL_FWD
    o.backward(_t(d["grad_o"]))
    for t, n in ((q, "dq"), (k, "dk"), (v, "dv")):
        # atol: gradients that are exactly zero in exact arithmetic (N=1 masked) come out as fp32 noise
        assert rel_err(t.grad.cpu().numpy(), d[n], atol=2e-2) < TOL_BWD, n


@pytest.mark.parametrize("path", ["quadratic", "recurrent", "mfma"])
@pytest.mark.parametrize("name", [n for n in golden_names("fm_") if n.endswith("p1_masked")] +
                         ["opt_tensors_normalized_p1", "opt_normalize_term_3_p1"])
def test_golden_forward_each_kernel_family(name, path):
    from attention_mechanisms.fastmax import fastmax
    from fastmax_experiments_amd import _lib, ops
    d, meta = load_golden(name)
    q, k, v = (_t(d[n]) for n in "qkv")
    _force(path)
    if path == "mfma" and ops.selected_path(q, k, 1, True) != _lib.PATH_MFMA:
        pytest.skip("matrix-core kernel does not cover this head size")
    o = fastmax(q, k, v, mask=True, p=1, **_kw(meta))
    assert rel_err(o.cpu().numpy(), d["o"]) < TOL_FWD


@pytest.mark.parametrize("path", ["quadratic", "quadratic_mfma"])
@pytest.mark.parametrize("name", [n for n in golden_names("fm_") if "N1D" not in n and "N7D" not in n] +
                         ["opt_tensors_normalized_p2", "opt_normalize_term_3_p2_unmasked"])
def test_golden_forward_quadratic_families(name, path):
    """every p / mask combination through the vector-ALU and the matrix-core tile kernels"""
    from attention_mechanisms.fastmax import fastmax
    d, meta = load_golden(name)
    _force(path)
    o = fastmax(*(_t(d[n]) for n in "qkv"), mask=meta["mask"], p=meta["p"], **_kw(meta))
    assert rel_err(o.cpu().numpy(), d["o"]) < TOL_FWD


@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 8e-3), (torch.float16, 2e-3), (torch.float32, TOL_FWD)])
@pytest.mark.parametrize("shape,p,mask", [((2, 3, 200, 64), 2, True), ((1, 2, 130, 128), 2, True), ((1, 2, 96, 32), 2, True),
                                          ((2, 2, 100, 64), 1, False), ((1, 2, 257, 128), 2, False), ((1, 2, 70, 80), 2, True)])
def test_matrix_core_tiles_dtypes_and_head_sizes(shape, p, mask, dt, tol):
    from attention_mechanisms.fastmax import fastmax
    from fastmax_experiments_amd import _lib, ops
    from oracle import c_oracle
    g = torch.Generator().manual_seed(shape[2])
    q, k, v = (torch.randn(shape, generator=g).to(dt) for _ in range(3))
    qq, kk, vv = q.cuda(), k.cuda(), v.cuda()
    assert ops.selected_path(qq, kk, p, mask) == _lib.PATH_QUADRATIC_MFMA
    o = fastmax(qq, kk, vv, mask=mask, p=p)
    ro, _ = c_oracle.fwd(q.float().numpy(), k.float().numpy(), v.float().numpy(), mask=mask, p=p)
    assert rel_err(o.float().cpu().numpy(), ro) < tol


def test_decode_shapes_matrix_core_tiles():
    # N_q != N_k, unmasked (KV-cache prefill/decode shapes, model.py:427-430,464-466)
    from attention_mechanisms.fastmax import fastmax
    from oracle import fastmax_oracle as orc
    g = torch.Generator().manual_seed(9)
    # (1, 1000), (3, 300), (15, 4096): a few new tokens against a long cache -- matrix-core tiles with idle query rows
    for nq, nk in ((16, 100), (64, 64), (100, 333), (1, 1000), (3, 300), (15, 4096), (1, 40)):
        q, k, v = torch.randn(2, 2, nq, 64, generator=g), torch.randn(2, 2, nk, 64, generator=g), torch.randn(2, 2, nk, 64, generator=g)
        for p in (1, 2):
            o = fastmax(q.cuda(), k.cuda(), v.cuda(), mask=False, p=p)
            ro, _ = orc.fastmax_fwd_dense(q.numpy(), k.numpy(), v.numpy(), mask=False, p=p)
            # unmasked g carries the constant N_q (fastmax.py:271): with one or three queries g = N_q + a q.ksum can be close
            # to zero and the quotient is ill-conditioned, whatever computes it
            assert rel_err(o.cpu().numpy(), ro) < (TOL_FWD if nq >= 16 else 2e-3)
    for dt, tol in ((torch.bfloat16, 8e-3), (torch.float16, 2e-3)):          # 16-bit single-token decode, D = 128 and 256
        for D in (128, 256):
            q, k, v = (torch.randn(1, 4, n, D, generator=g).to(dt) for n in (1, 2000, 2000))
            o = fastmax(q.cuda(), k.cuda(), v.cuda(), mask=False, p=2)
            ro, _ = orc.fastmax_fwd_dense(q.float().numpy(), k.float().numpy(), v.float().numpy(), mask=False, p=2)
            assert rel_err(o.float().cpu().numpy(), ro) < tol, D


@pytest.mark.parametrize("dt,tf,tb", [(torch.float32, TOL_FWD, TOL_BWD), (torch.bfloat16, 8e-3, 2.5e-2), (torch.float16, 2e-3, 5e-3)])
@pytest.mark.parametrize("nq,nk,D", [(700, 1100, 64), (64, 4096, 128), (2048, 2048, 64), (300, 520, 48), (1000, 640, 128)])
def test_unmasked_first_order_in_linear_time(nq, nk, D, dt, tf, tb):
    """mask=False, p=1 at sizes where the O(N_q N_k) tiles lose: o_i = (S1 + a S2^T q_i) / (g0 + a q_i.ksum) from ONE pass over K, V
    (the sequence-split state kernel, all segments) and a D x D product per query row -- what the reference's KV-cache inference
    calls (model.py:460-487: the whole prompt against the cache, mask=False).  Forward against the dense fp64 oracle for both
    denominators' constants (N_q in fastmax.py:271, N_k in fastmax_hack.py:21), and the gradients -- from totals as well
    (dQ_i = a w_i S2 G_i + a e_i ksum, dK_j = a (R2 v_j + rq), dV_j = R1 + a R2^T k_j; fp32 / fp16 above D = 64: tile kernels fed with
    this forward's o and g) -- against the C oracle."""
    from attention_mechanisms.fastmax import fastmax
    from attention_mechanisms.fastmax_hack import fastmax_hack
    from fastmax_experiments_amd import _lib, ops
    from oracle import c_oracle, fastmax_oracle as orc
    g = torch.Generator().manual_seed(nq + nk + D)
    q, go = (torch.randn(2, 3, nq, D, generator=g).to(dt) for _ in range(2))
    k, v = (torch.randn(2, 3, nk, D, generator=g).to(dt) for _ in range(2))
    qq, kk, vv = (t.cuda().requires_grad_(True) for t in (q, k, v))
    assert ops.selected_path(qq, kk, 1, False) == _lib.PATH_MFMA
    o = fastmax(qq, kk, vv, mask=False, p=1)
    ro, _ = orc.fastmax_fwd_dense(q.float().numpy(), k.float().numpy(), v.float().numpy(), mask=False, p=1)
    assert rel_err(o.detach().float().cpu().numpy(), ro) < (tf if o.dtype == dt else max(tf, TOL_FWD))
    o.backward(go.cuda().to(o.dtype))
    e = c_oracle.bwd(q.float().numpy(), k.float().numpy(), v.float().numpy(), go.float().numpy(), mask=False, p=1)
    for t, rr, n in zip((qq, kk, vv), e, ("dq", "dk", "dv")):
        assert rel_err(t.grad.float().cpu().numpy(), rr, atol=2e-2) < tb, n
    with torch.no_grad():
        oh = fastmax_hack(q.cuda(), k.cuda(), v.cuda(), p=1, mask=False)
    rh = orc.linearmax_fwd(q.double().numpy(), k.double().numpy(), v.double().numpy(), p=1, mask=False)
    assert rel_err(oh.float().cpu().numpy(), np.asarray(rh)) < max(4 * tf, 2e-3)


def test_c1_baseline_config():
    from attention_mechanisms.fastmax import fastmax
    d, _ = load_golden("c1_fastmax_p1_masked_fp32")
    o = fastmax(*(_t(d[n]) for n in "qkv"))
    assert rel_err(o.cpu().numpy(), d["o"]) < TOL_FWD


def test_stress_large_scores_recorded():
    # N(0,16) scores: the reference's own fp32 run is ill-conditioned here (1/g amplification);
    # recorded with a loose bound, as SURVEY 8c prescribes
    from attention_mechanisms.fastmax import fastmax
    for name in golden_names("stress_"):
        d, meta = load_golden(name)
        o = fastmax(*(_t(d[n]) for n in "qkv"), mask=True, p=meta["p"])
        assert rel_err(o.cpu().numpy(), d["o"]) < 5e-2


def test_noncontiguous_gqa_inputs():
    from attention_mechanisms.fastmax import fastmax
    d, _ = load_golden("noncontig_gqa_p1")
    B, H, N, D = d["q"].shape
    q = _t(d["q"]).transpose(1, 2).contiguous().transpose(1, 2)       # strided view
    kg = _t(d["k"][:, ::3])                                            # one row per KV group
    k = kg[:, :, None].expand(B, H // 3, 3, N, D).reshape(B, H, N, D)
    v = _t(d["v"])
    assert not q.is_contiguous()
    o = fastmax(q, k, v)
    assert o.is_contiguous() and rel_err(o.cpu().numpy(), d["o"]) < TOL_FWD
    # a strided view (every other token of a padded buffer) is consumed in place through the strides
    qpad = torch.zeros(B, H, 2 * N, D, device="cuda")
    qpad[:, :, ::2] = _t(d["q"])
    qs = qpad[:, :, ::2]
    assert qs.stride(2) == 2 * D
    o2 = fastmax(qs, k, v)
    assert rel_err(o2.cpu().numpy(), d["o"]) < TOL_FWD


@pytest.mark.parametrize("name", golden_names("decode_"))
def test_unmasked_nq_ne_nk(name):
    from attention_mechanisms.fastmax import fastmax
    d, meta = load_golden(name)
    o = fastmax(*(_t(d[n]) for n in "qkv"), mask=False, p=meta["p"])
    assert o.shape == d["o"].shape and rel_err(o.cpu().numpy(), d["o"]) < TOL_FWD


@pytest.mark.parametrize("name", ["hack_masked_p1", "hack_masked_p2", "hack_masked_p1_D128", "hack_unmasked",
                                  "hack_unmasked_Nq4_Nk16"])
def test_linearmax_forward(name):
    from attention_mechanisms.fastmax_hack import fastmax_hack
    d, meta = load_golden(name)
    o = fastmax_hack(*(_t(d[n]) for n in "qkv"), p=meta["p"], mask=meta["mask"])
    assert o.dtype == torch.float32 and rel_err(o.cpu().numpy(), d["o"]) < TOL_FWD


def test_linearmax_gradients():
    from attention_mechanisms.fastmax_hack import fastmax_hack
    d, _ = load_golden("hack_masked_p1_grads")
    q, k, v = (_t(d[n], grad=True) for n in "qkv")
    o = fastmax_hack(q, k, v, p=1, mask=True)
    assert rel_err(o.detach().cpu().numpy(), d["o"]) < TOL_FWD
    o.backward(_t(d["grad_o"]))
    for t, n in ((q, "dq"), (k, "dk"), (v, "dv")):
        assert rel_err(t.grad.cpu().numpy(), d[n]) < TOL_BWD, n


def test_normalize():
    from attention_mechanisms.fastmax import fastattention_einops
    d, _ = load_golden("normalize")
    qn, kn = fastattention_einops.normalize(_t(d["q"]), _t(d["k"]))
    assert rel_err(qn.cpu().numpy(), d["qn"]) < 1e-5 and rel_err(kn.cpu().numpy(), d["kn"]) < 1e-5


@pytest.mark.parametrize("name", golden_names("create_attn_"))
def test_create_attn(name, caplog):
    from attention_mechanisms.fastmax import fastmax
    d, meta = load_golden(name)
    o, a = fastmax(*(_t(d[n]) for n in "qkv"), mask=meta["mask"], p=meta["p"], create_attn=True)
    assert rel_err(a.cpu().numpy(), d["a"]) < 1e-5 and rel_err(o.cpu().numpy(), d["o"]) < 1e-5
    assert any("compute_attn = True" in r.message for r in caplog.records)


@pytest.mark.parametrize("dt", ["float32", "bfloat16", "float16"])
@pytest.mark.parametrize("mask", [True, False])
def test_dtype_rules(dt, mask):
    """Q1: masked keeps the input dtype, unmasked promotes bf16/fp16 to float32.  Values: low-precision
    kernels are judged against the fp64 ORACLE on the upcast inputs (the reference's own bf16 run
    accumulates its cumsums in bf16 and is 4.5e-3 off its fp32 self)."""
    from attention_mechanisms.fastmax import fastmax
    from oracle import fastmax_oracle as orc
    d, meta = load_golden(f"dtype_{dt}_{'masked' if mask else 'unmasked'}")
    tdt = getattr(torch, dt)
    q, k, v = (_t(d[n], tdt) for n in "qkv")
    o = fastmax(q, k, v, mask=mask, p=1)
    assert str(o.dtype) == meta["out_dtype"]
    ref, _ = orc.fastmax_fwd_factorized(*(t.float().cpu().numpy() for t in (q, k, v)), mask=mask, p=1)
    tol = 2e-4 if o.dtype == torch.float32 else (8e-3 if o.dtype == torch.bfloat16 else 2e-3)
    assert rel_err(o.float().cpu().numpy(), ref) < tol
    # and it stays in the neighbourhood of what the reference itself produced in that dtype
    assert rel_err(o.float().cpu().numpy(), d["o"]) < 2e-2


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_low_precision_backward(dt):
    from attention_mechanisms.fastmax import fastmax
    from oracle import c_oracle
    g = torch.Generator().manual_seed(3)
    q, k, v, go = (torch.randn(2, 2, 96, 64, generator=g).to(dt) for _ in range(4))
    for mask in (True, False):
        qq, kk, vv = (t.cuda().requires_grad_(True) for t in (q, k, v))
        o = fastmax(qq, kk, vv, mask=mask, p=2)
        o.backward(go.cuda().to(o.dtype))
        assert qq.grad.dtype == dt and kk.grad.dtype == dt and vv.grad.dtype == dt
        e = c_oracle.bwd(q.float().numpy(), k.float().numpy(), v.float().numpy(), go.float().numpy(), mask=mask, p=2)
        tol = 2e-2 if dt == torch.bfloat16 else 4e-3
        for t, r in zip((qq, kk, vv), e):
            assert rel_err(t.grad.float().cpu().numpy(), r) < tol


@pytest.mark.parametrize("dt,tol", [(torch.float32, TOL_BWD), (torch.bfloat16, 2e-2), (torch.float16, 4e-3)])
@pytest.mark.parametrize("shape,p,mask", [((2, 3, 200, 64), 2, True), ((1, 2, 130, 128), 2, True), ((1, 2, 96, 32), 1, True),
                                          ((2, 2, 100, 64), 1, False), ((1, 2, 257, 128), 2, False), ((1, 2, 70, 80), 2, True),
                                          ((1, 1, 5, 16), 2, True)])
def test_backward_matrix_core_vs_vector_alu_vs_oracle(shape, p, mask, dt, tol):
    """dQ, dK, dV: matrix-core tiles and the vector-ALU family against the C oracle (fp64)"""
    from attention_mechanisms.fastmax import fastmax
    from oracle import c_oracle
    g = torch.Generator().manual_seed(shape[2] + p)
    q, k, v, go = (torch.randn(shape, generator=g).to(dt) for _ in range(4))
    e = c_oracle.bwd(q.float().numpy(), k.float().numpy(), v.float().numpy(), go.float().numpy(), mask=mask, p=p)
    for path in ("auto", "quadratic"):
        _force(path)
        qq, kk, vv = (t.cuda().requires_grad_(True) for t in (q, k, v))
        o = fastmax(qq, kk, vv, mask=mask, p=p)
        o.backward(go.cuda().to(o.dtype))
        for t, rr, n in zip((qq, kk, vv), e, ("dq", "dk", "dv")):
            assert t.grad.dtype == dt
            assert rel_err(t.grad.float().cpu().numpy(), rr) < tol, (path, n)


@pytest.mark.parametrize("dt,tolf,tolb", [(torch.float32, TOL_FWD, TOL_BWD), (torch.bfloat16, 8e-3, 2e-2), (torch.float16, 2e-3, 4e-3)])
@pytest.mark.parametrize("shape,p,mask", [((1, 2, 300, 256), 2, True), ((2, 1, 130, 192), 2, True), ((1, 2, 257, 136), 1, True),
                                          ((1, 2, 200, 256), 2, False), ((1, 1, 70, 160), 1, False)])
def test_head_sizes_above_128(shape, p, mask, dt, tolf, tolb):
    """head sizes 136 .. 256 (lit_gpt/config.py: pythia-1b and Gemma-2b 256, stablelm-tuned-alpha-3b and Gemma-7b 192): the tile
    kernels with 256-column images -- matrix cores for the forward (every dtype) and the bf16 backward, the vector-ALU tiles for
    the fp32 / fp16 backward and on request -- against the C oracle (fp64), forward and gradients"""
    from attention_mechanisms.fastmax import fastmax
    from fastmax_experiments_amd import _lib, ops
    from oracle import c_oracle
    g = torch.Generator().manual_seed(shape[2] + shape[3] + p)
    q, k, v, go = (torch.randn(shape, generator=g).to(dt) for _ in range(4))
    qn, kn, vn, gn = (t.float().numpy() for t in (q, k, v, go))
    ro, _ = c_oracle.fwd(qn, kn, vn, mask=mask, p=p)
    e = c_oracle.bwd(qn, kn, vn, gn, mask=mask, p=p)
    assert ops.selected_path(q.cuda(), k.cuda(), p, mask) == _lib.PATH_QUADRATIC_MFMA
    for path in ("auto", "quadratic"):
        _force(path)
        qq, kk, vv = (t.cuda().requires_grad_(True) for t in (q, k, v))
        o = fastmax(qq, kk, vv, mask=mask, p=p)
        assert rel_err(o.detach().float().cpu().numpy(), ro) < tolf, path
        o.backward(go.cuda().to(o.dtype))
        for t, rr, n in zip((qq, kk, vv), e, ("dq", "dk", "dv")):
            assert t.grad.dtype == dt
            assert rel_err(t.grad.float().cpu().numpy(), rr) < tolb, (path, n)
    _force("auto")


@pytest.mark.parametrize("dt,tol", [(torch.float32, TOL_FWD), (torch.bfloat16, 8e-3), (torch.float16, 2e-3)])
@pytest.mark.parametrize("shape", [(2, 3, 200, 64), (1, 2, 1000, 32), (1, 2, 130, 16), (1, 2, 257, 48), (2, 2, 65, 64),
                                   (1, 2, 300, 128), (1, 2, 129, 96)])
def test_linear_time_matrix_core_kernel_dtypes_and_head_sizes(shape, dt, tol):
    """p=1 masked through the generic chunked-scan kernel (padded head sizes, 16-bit inputs, D=128 for bf16)"""
    from attention_mechanisms.fastmax import fastmax
    from fastmax_experiments_amd import _lib, ops
    from oracle import c_oracle
    g = torch.Generator().manual_seed(shape[2] + shape[3])
    q, k, v = (torch.randn(shape, generator=g).to(dt) for _ in range(3))
    qq, kk, vv = q.cuda(), k.cuda(), v.cuda()
    # D <= 64: generic / bf16 kernels; 64 < D <= 128: the eight-wave kernels (bf16, and fp32 / fp16 with two-part operands)
    want = _lib.PATH_MFMA
    assert ops.selected_path(qq, kk, 1, True) == want
    o = fastmax(qq, kk, vv)
    assert o.dtype == dt
    ro, _ = c_oracle.fwd(q.float().numpy(), k.float().numpy(), v.float().numpy())
    assert rel_err(o.float().cpu().numpy(), ro) < tol


@pytest.mark.parametrize("dt,tol", [(torch.float32, 3e-4), (torch.bfloat16, 1.5e-2), (torch.float16, 3e-3)])
@pytest.mark.parametrize("shape", [(2, 3, 200, 64), (1, 4, 1024, 32), (1, 2, 333, 128), (1, 2, 70, 80), (1, 2, 200, 256), (1, 2, 130, 192)])
def test_linearmax_fused_prologue(shape, dt, tol):
    """fastmax_hack masked, forward only: prologue fused into the kernel (where covered) vs the fp64 oracle"""
    from attention_mechanisms.fastmax_hack import fastmax_hack
    from oracle import fastmax_oracle as orc
    g = torch.Generator().manual_seed(shape[2])
    q, k, v = (torch.randn(shape, generator=g).to(dt) for _ in range(3))
    with torch.no_grad():
        o = fastmax_hack(q.cuda(), k.cuda(), v.cuda(), p=1, mask=True)
    assert o.dtype == dt
    ro = orc.linearmax_fwd(q.float().numpy(), k.float().numpy(), v.float().numpy(), chunk=64)
    assert rel_err(o.float().cpu().numpy(), ro) < tol
    # the autograd (unfused) route computes the same function
    qq = q.cuda().requires_grad_(True)
    o2 = fastmax_hack(qq, k.cuda(), v.cuda(), p=1, mask=True)
    assert rel_err(o2.detach().float().cpu().numpy(), ro) < tol
    o2.float().sum().backward()
    assert qq.grad is not None and qq.grad.dtype == dt and torch.isfinite(qq.grad).all()


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(2, 3, 700, 64), (1, 5, 257, 128), (1, 2, 70, 80), (3, 1, 4, 16)])
def test_paired_prologue_statistics_match_the_single_tensor_pass(shape, dt):
    """fastmax_hip_normalize_stats2 (q and k in two launches in all) == two fastmax_hip_normalize_stats passes, bit for bit,
    also on strided (transposed-storage) views; and against the definition max_n ||x_n - mean(x_n)|| (fastmax_hack.py:38-43)"""
    from fastmax_experiments_amd import ops
    g = torch.Generator().manual_seed(shape[2] + shape[3])
    q = torch.randn(shape, generator=g).to(dt).cuda()
    k = (torch.randn(shape, generator=g) * 1.7).to(dt).cuda()
    kt = k.transpose(1, 2).contiguous().transpose(1, 2)                   # same values, (B, N, H, D) storage
    for kk in (k, kt):
        qi, ki = ops.normalize_stats_pair(q, kk)
        assert torch.equal(qi, ops.normalize_stats(q)) and torch.equal(ki, ops.normalize_stats(kk))
    xc = k.float() - k.float().mean(-1, keepdim=True)
    want = 1.0 / xc.norm(dim=-1).amax(-1)
    assert torch.allclose(ki, want, rtol=2e-6)


@pytest.mark.parametrize("dt,tol", [(torch.float32, TOL_FWD), (torch.bfloat16, 8e-3)])
@pytest.mark.parametrize("shape", [(1, 2, 4096, 64), (1, 3, 1500, 64), (1, 1, 2048, 32), (1, 2, 2100, 128), (2, 8, 1024, 64)])
def test_sequence_split_for_few_heads(shape, dt, tol):
    """few heads -> the sequence is cut into segments (state kernel + prefix + main kernel): same result"""
    from attention_mechanisms.fastmax import fastmax
    from attention_mechanisms.fastmax_hack import fastmax_hack
    from oracle import c_oracle, fastmax_oracle as orc
    g = torch.Generator().manual_seed(shape[2])
    q, k, v = (torch.randn(shape, generator=g).to(dt) for _ in range(3))
    o = fastmax(q.cuda(), k.cuda(), v.cuda())
    ro, _ = c_oracle.fwd(q.float().numpy(), k.float().numpy(), v.float().numpy())
    assert rel_err(o.float().cpu().numpy(), ro) < tol
    with torch.no_grad():
        oh = fastmax_hack(q.cuda(), k.cuda(), v.cuda())
    rh = orc.linearmax_fwd(q.float().numpy(), k.float().numpy(), v.float().numpy(), chunk=64)
    assert rel_err(oh.float().cpu().numpy(), rh) < 2 * tol


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2 * TOL_FWD), (torch.bfloat16, 1.6e-2), (torch.float16, 3e-3)])
@pytest.mark.parametrize("shape", [(1, 2, 4096, 64), (1, 3, 1500, 48), (1, 2, 2100, 128), (1, 1, 8192, 128), (2, 8, 1024, 64), (2, 8, 320, 64)])
def test_linearmax_statistics_ride_on_the_state_pass(shape, dt, tol):
    """fastmax_hip_linearmax_forward_auto: with the sequence split the prologue statistics come out of the split's state pass
    (K's words from the blocks that build the states, Q's from extra blocks of the same launch; the records are scaled by the
    prefix pass).  The statistics it leaves == the definition (fastmax_hack.py:38-43), the output == the fp64 oracle, also for a
    K of another magnitude in (B, N, H, D) storage; the last shape is too short for the split (paired statistics pass)."""
    from fastmax_experiments_amd import ops
    from oracle import fastmax_oracle as orc
    g = torch.Generator().manual_seed(shape[2] + shape[3])
    q = torch.randn(shape, generator=g).to(dt)
    k = (torch.randn(shape, generator=g) * 2.3 + 0.4).to(dt)
    v = torch.randn(shape, generator=g).to(dt)
    kd = k.cuda().transpose(1, 2).contiguous().transpose(1, 2)
    o, qi, ki = ops.linearmax_forward_fused(q.cuda(), kd, v.cuda(), return_stats=True)
    for x, inv in ((q, qi), (k, ki)):
        xc = x.double() - x.double().mean(-1, keepdim=True)
        want = (1.0 / xc.norm(dim=-1).amax(-1)).reshape(-1)
        assert torch.allclose(inv.double().cpu(), want, rtol=3e-6, atol=0.0)
    ro = orc.linearmax_fwd(q.float().numpy(), k.float().numpy(), v.float().numpy(), chunk=64)
    assert o.dtype == dt and rel_err(o.float().cpu().numpy(), ro) < tol


@pytest.mark.parametrize("dt,tol", [(torch.float32, TOL_BWD), (torch.bfloat16, 2e-2), (torch.float16, 4e-3)])
@pytest.mark.parametrize("shape", [(2, 3, 640, 64), (1, 2, 1000, 32), (1, 2, 513, 48), (1, 1, 2048, 64), (1, 2, 777, 128), (1, 1, 2100, 128),
                                   (2, 2, 1024, 96)])
def test_linear_time_backward(shape, dt, tol):
    """p=1 masked backward by forward / reverse scans with carried state vs the C oracle and vs the tile kernels"""
    from attention_mechanisms.fastmax import fastmax
    from oracle import c_oracle
    g = torch.Generator().manual_seed(shape[2])
    q, k, v, go = (torch.randn(shape, generator=g).to(dt) for _ in range(4))
    e = c_oracle.bwd(q.float().numpy(), k.float().numpy(), v.float().numpy(), go.float().numpy(), mask=True, p=1)
    grads = {}
    for path in ("auto", "quadratic_mfma"):
        _force(path)
        qq, kk, vv = (t.cuda().requires_grad_(True) for t in (q, k, v))
        o = fastmax(qq, kk, vv)
        o.backward(go.cuda())
        grads[path] = [t.grad.float().cpu().numpy() for t in (qq, kk, vv)]
        for gr, rr, n in zip(grads[path], e, ("dq", "dk", "dv")):
            assert rel_err(gr, rr) < tol, (path, n)


def test_cpu_tensors_round_trip_like_model_py():
    # lit_gpt/model.py:482-486 hands CPU tensors over and calls .cuda() on the result
    from attention_mechanisms.fastmax import fastmax
    d, _ = load_golden("fm_B1H2N64D64_p2_masked")
    q, k, v = (torch.from_numpy(d[n]) for n in "qkv")
    o = fastmax(q, k, v, p=2, mask=True)
    assert o.device.type == "cpu" and rel_err(o.numpy(), d["o"]) < TOL_FWD
    assert o.cuda().is_cuda


def test_float64_inputs_keep_dtype():
    from attention_mechanisms.fastmax import fastmax
    d, _ = load_golden("fm_B1H2N64D32_p1_masked")
    o = fastmax(*(_t(d[n], torch.float64) for n in "qkv"))
    assert o.dtype == torch.float64 and rel_err(o.cpu().numpy(), d["o"]) < TOL_FWD


def test_bad_p_raises():
    from attention_mechanisms.fastmax import fastmax
    q = torch.zeros(1, 1, 4, 8, device="cuda")
    with pytest.raises(ValueError):
        fastmax(q, q, q, p=3)


@pytest.mark.parametrize("shape,p,mask", [((2, 4, 1024, 64), 1, True), ((1, 3, 1000, 64), 1, True),
                                          ((1, 2, 777, 128), 1, True), ((2, 2, 513, 32), 1, True),
                                          ((1, 2, 300, 80), 1, True), ((1, 2, 512, 64), 2, True),
                                          ((1, 2, 384, 64), 2, False), ((1, 2, 300, 48), 1, False)])
def test_medium_sizes_vs_c_oracle(shape, p, mask):
    from attention_mechanisms.fastmax import fastmax
    from oracle import c_oracle
    g = torch.Generator().manual_seed(11)
    q, k, v, go = (torch.randn(shape, generator=g) for _ in range(4))
    qq, kk, vv = (t.cuda().requires_grad_(True) for t in (q, k, v))
    o = fastmax(qq, kk, vv, mask=mask, p=p)
    ro, _ = c_oracle.fwd(q.numpy(), k.numpy(), v.numpy(), mask=mask, p=p)
    assert rel_err(o.detach().cpu().numpy(), ro) < TOL_FWD
    o.backward(go.cuda())
    e = c_oracle.bwd(q.numpy(), k.numpy(), v.numpy(), go.numpy(), mask=mask, p=p)
    for t, r, n in zip((qq, kk, vv), e, ("dq", "dk", "dv")):
        assert rel_err(t.grad.cpu().numpy(), r) < TOL_BWD, n


@pytest.mark.parametrize("path", ["recurrent", "mfma"])
def test_kernel_families_agree_on_ragged_lengths(path):
    from attention_mechanisms.fastmax import fastmax
    from fastmax_experiments_amd import _lib, ops
    from oracle import c_oracle
    g = torch.Generator().manual_seed(5)
    for N in (1, 2, 15, 16, 17, 63, 64, 65, 127, 129, 200):
        q, k, v = (torch.randn(2, 3, N, 64, generator=g) for _ in range(3))
        _force(path)
        if path == "mfma" and ops.selected_path(q.cuda(), k.cuda(), 1, True) != _lib.PATH_MFMA:
            pytest.skip("matrix-core kernel not available for this shape")
        o = fastmax(q.cuda(), k.cuda(), v.cuda())
        ro, _ = c_oracle.fwd(q.numpy(), k.numpy(), v.numpy())
        assert rel_err(o.cpu().numpy(), ro) < TOL_FWD, N


def test_full_baseline_shape_properties():
    """BASELINE shape (16,32,4096,64) fp32, p=1 masked: sampled heads against the C oracle plus
    size-independent properties -- rows of the implied attention matrix sum to one (v = ones -> o = 1),
    linearity in V, and causality (changing tokens >= t leaves outputs < t untouched, bit for bit)."""
    from attention_mechanisms.fastmax import fastmax
    from oracle import c_oracle
    B, H, N, D = 16, 32, 4096, 64
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v = (torch.randn(B, H, N, D, device="cuda", generator=g) for _ in range(3))
    o = fastmax(q, k, v)
    for (b, h) in ((0, 0), (7, 13), (15, 31)):
        ro, _ = c_oracle.fwd(*(t[b:b + 1, h:h + 1].cpu().numpy() for t in (q, k, v)))
        assert rel_err(o[b:b + 1, h:h + 1].cpu().numpy(), ro) < TOL_FWD
    ones = fastmax(q, k, torch.ones_like(v))
    assert float((ones - 1).abs().max()) < 1e-4
    v2 = torch.randn(B, H, N, D, device="cuda", generator=g)
    o2 = fastmax(q, k, v2)
    o3 = fastmax(q, k, 0.5 * v - 2.0 * v2)
    assert float((o3 - (0.5 * o - 2.0 * o2)).abs().max()) < 1e-4 * float(o.abs().max() + 2 * o2.abs().max())
    t = 2049
    q2, k2, v3 = q.clone(), k.clone(), v.clone()
    q2[:, :, t:], k2[:, :, t:], v3[:, :, t:] = 1.5, -0.5, 3.0
    o4 = fastmax(q2, k2, v3)
    assert torch.equal(o4[:, :, :t], o[:, :, :t])
    assert not torch.equal(o4[:, :, t:], o[:, :, t:])


def test_config4_full_size_second_order_d128():
    """BASELINE config 4 head shape at its full length: fastmax(p=2) -- what lit_gpt/model.py:485 calls -- on
    (2,32,4096,128) bf16, forward + backward.  Sampled heads against the dense fp64 known-answer form of the
    oracle (fastmax.py:336-381 + 103 and its derivative) on the upcast inputs, plus size-independent properties:
    rows of the implied attention matrix sum to one, causality bit for bit (outputs, and nothing upstream of a
    gradient cut-off moves), gradient of a constant-one value tensor sums to zero in q and k."""
    from attention_mechanisms.fastmax import fastmax
    from oracle import fastmax_oracle as orc
    B, H, N, D = 2, 32, 4096, 128
    g = torch.Generator(device="cuda").manual_seed(4)
    q, k, v, go = (torch.randn(B, H, N, D, device="cuda", generator=g).to(torch.bfloat16) for _ in range(4))
    q.requires_grad_(True), k.requires_grad_(True), v.requires_grad_(True)
    o = fastmax(q, k, v, mask=True, p=2)
    assert o.shape == (B, H, N, D) and o.dtype == torch.bfloat16 and o.is_contiguous()
    o.backward(go)
    for (b, h) in ((0, 0), (1, 31)):
        qn, kn, vn, gn = (t[b:b + 1, h:h + 1].detach().float().cpu().numpy() for t in (q, k, v, go))
        ro, _ = orc.fastmax_fwd_dense(qn, kn, vn, mask=True, p=2)
        assert rel_err(o[b:b + 1, h:h + 1].detach().float().cpu().numpy(), ro) < 8e-3
        for t, r, n in zip((q, k, v), orc.fastmax_bwd_dense(qn, kn, vn, gn, mask=True, p=2), "qkv"):
            assert rel_err(t.grad[b:b + 1, h:h + 1].float().cpu().numpy(), r) < 2.5e-2, n
    with torch.no_grad():
        ones = fastmax(q, k, torch.ones_like(v), mask=True, p=2)
        assert float((ones.float() - 1).abs().max()) < 8e-3                     # one bf16 rounding of 1 +- eps
        t = 2049
        q2, k2, v2 = q.detach().clone(), k.detach().clone(), v.detach().clone()
        q2[:, :, t:], k2[:, :, t:], v2[:, :, t:] = 1.5, -0.5, 3.0
        o2 = fastmax(q2, k2, v2, mask=True, p=2)
        assert torch.equal(o2[:, :, :t], o.detach()[:, :, :t]) and not torch.equal(o2[:, :, t:], o.detach()[:, :, t:])
    # a gradient that is zero from token t on leaves dK, dV of the tokens >= t exactly zero (nothing attends backwards)
    go2 = go.clone()
    go2[:, :, t:] = 0
    qq, kk, vv = (x.detach().clone().requires_grad_(True) for x in (q, k, v))
    fastmax(qq, kk, vv, mask=True, p=2).backward(go2)
    assert float(kk.grad[:, :, t:].abs().max()) == 0 and float(vv.grad[:, :, t:].abs().max()) == 0
    assert float(qq.grad[:, :, t:].abs().max()) == 0 and float(qq.grad[:, :, :t].abs().max()) > 0


def _prologue_backward_fp64(x, gy):
    """d/dx of y = (x - mean_D x) / max_n ||x_n - mean_D x_n|| (fastmax.py:326-334), one head (N,D), fp64: the chain rule
    through the oracle's normalize -- only the row that attains the max-norm carries the dL/dM term."""
    xc = x - x.mean(-1, keepdims=True)
    nrm = np.sqrt((xc * xc).sum(-1))
    ns = int(nrm.argmax())
    M = nrm[ns]
    gxc = gy / M
    gxc[ns] -= (gy * xc).sum() / (M * M) * xc[ns] / M
    return gxc - gxc.mean(-1, keepdims=True)


def test_config5_full_size_linearmax_16k():
    """BASELINE config 5 at its full length: fastmax_hack (linearmax, fastmax_hack.py:36-60) on (1,32,16384,128) bf16,
    forward + backward (the O(N) causal scan over 256 chunks, sequence split included).  Sampled heads against the C
    oracle (prologue + first-order scan with nt=1, and the chain rule through the prologue for dq, dk) on the upcast
    inputs, plus rows-sum-to-one and bit-exact causality."""
    from attention_mechanisms.fastmax_hack import fastmax_hack
    from oracle import c_oracle, fastmax_oracle as orc
    B, H, N, D = 1, 32, 16384, 128
    g = torch.Generator(device="cuda").manual_seed(5)
    q, k, v, go = (torch.randn(B, H, N, D, device="cuda", generator=g).to(torch.bfloat16) for _ in range(4))
    q.requires_grad_(True), k.requires_grad_(True), v.requires_grad_(True)
    o = fastmax_hack(q, k, v, p=1, mask=True)
    assert o.shape == (B, H, N, D) and o.dtype == torch.bfloat16
    o.backward(go)
    for h in (0, 17, 31):
        qn, kn, vn, gn = (t[:, h:h + 1].detach().float().cpu().numpy() for t in (q, k, v, go))
        qq, kk = orc.normalize_qk(qn, kn)
        qq32, kk32 = qq.astype(np.float32), kk.astype(np.float32)              # what the C oracle takes
        ro, _ = c_oracle.fwd(qq32, kk32, vn, mask=True, nt=1.0, p=1)
        assert rel_err(o[:, h:h + 1].detach().float().cpu().numpy(), ro) < 8e-3, h
        dqn, dkn, dv = c_oracle.bwd(qq32, kk32, vn, gn, mask=True, nt=1.0, p=1)
        dq = _prologue_backward_fp64(qn[0, 0].astype(np.float64), np.asarray(dqn, dtype=np.float64)[0, 0])
        dk = _prologue_backward_fp64(kn[0, 0].astype(np.float64), np.asarray(dkn, dtype=np.float64)[0, 0])
        assert rel_err(v.grad[0, h].float().cpu().numpy(), dv[0, 0]) < 2.5e-2, h
        assert rel_err(q.grad[0, h].float().cpu().numpy(), dq) < 2.5e-2, h
        assert rel_err(k.grad[0, h].float().cpu().numpy(), dk) < 2.5e-2, h
    with torch.no_grad():
        ones = fastmax_hack(q, k, torch.ones_like(v), p=1, mask=True)
        assert float((ones.float() - 1).abs().max()) < 8e-3
        # causality: V of the tokens >= t does not reach outputs < t (q, k are left alone: the prologue's max-norm is global)
        t = 8193
        # (both sides without autograd: the inference route fuses the prologue into the scan kernel and keeps the
        # normalised q, k in fp32, the training route rounds them to bf16 -- not the same bits)
        o1 = fastmax_hack(q, k, v, p=1, mask=True)
        assert rel_err(o1.float().cpu().numpy(), o.detach().float().cpu().numpy()) < 8e-3
        v2 = v.detach().clone()
        v2[:, :, t:] = 3.0
        o2 = fastmax_hack(q, k, v2, p=1, mask=True)
        assert torch.equal(o2[:, :, :t], o1[:, :, :t]) and not torch.equal(o2[:, :, t:], o1[:, :, t:])


@pytest.mark.parametrize("dt,tol", [(torch.float32, TOL_FWD), (torch.bfloat16, 8e-3)])
@pytest.mark.parametrize("B,H,T,D", [(2, 3, 200, 64), (1, 2, 70, 32), (1, 2, 130, 128)])
def test_decode_state_cache_matches_masked_forward(B, H, T, D, dt, tol):
    """opt-in O(D^2)-per-token decode: prefill + 6 single-token steps == rows of the masked forward over T+6 tokens"""
    from fastmax_experiments_amd.decode import FastmaxDecodeState
    from oracle import c_oracle
    if D == 128 and dt == torch.float32:
        pytest.skip("fp32 D=128 state kernel not built")
    g = torch.Generator().manual_seed(T)
    q, k, v = (torch.randn(B, H, T + 6, D, generator=g).to(dt) for _ in range(3))
    ref, _ = c_oracle.fwd(q.float().numpy(), k.float().numpy(), v.float().numpy())
    st = FastmaxDecodeState(B, H, D, "cuda")
    qc, kc, vc = q.cuda(), k.cuda(), v.cuda()
    o = st.prefill(qc[:, :, :T], kc[:, :, :T], vc[:, :, :T])
    assert rel_err(o.float().cpu().numpy(), ref[:, :, :T]) < tol
    for t in range(T, T + 6):
        ot = st.step(qc[:, :, t:t + 1], kc[:, :, t:t + 1], vc[:, :, t:t + 1])
        assert ot.shape == (B, H, 1, D) and ot.dtype == dt
        assert rel_err(ot.float().cpu().numpy(), ref[:, :, t:t + 1], atol=float(np.abs(ref).max())) < tol, t
    assert st.count == T + 6


def test_randomised_shapes_against_oracle():
    """seeded sweep over shapes / dtypes / p / mask (ragged lengths, padded head sizes, few and many heads):
    forward and backward through whatever kernel family the dispatcher picks, against the C oracle"""
    from attention_mechanisms.fastmax import fastmax
    from oracle import c_oracle
    rng = np.random.default_rng(2024)
    dts = [(torch.float32, TOL_FWD, TOL_BWD), (torch.bfloat16, 8e-3, 2.5e-2), (torch.float16, 2e-3, 5e-3)]
    for trial in range(36):
        B, H = int(rng.integers(1, 4)), int(rng.integers(1, 6))
        N = int(rng.choice([1, 3, 17, 63, 64, 65, 100, 191, 256, 333, 520, 777, 1100]))
        D = int(rng.choice([8, 16, 24, 32, 40, 64, 72, 96, 128]))
        p = int(rng.integers(1, 3))
        mask = bool(rng.integers(0, 2)) or N < 2
        dt, tf, tb = dts[trial % 3]
        g = torch.Generator().manual_seed(1000 + trial)
        q, k, v, go = (torch.randn(B, H, N, D, generator=g).to(dt) for _ in range(4))
        qq, kk, vv = (t.cuda().requires_grad_(True) for t in (q, k, v))
        o = fastmax(qq, kk, vv, mask=mask, p=p)
        ro, _ = c_oracle.fwd(q.float().numpy(), k.float().numpy(), v.float().numpy(), mask=mask, p=p)
        tag = (trial, B, H, N, D, p, mask, str(dt))
        tol_f = tf if o.dtype != torch.float32 or dt == torch.float32 else TOL_FWD
        assert rel_err(o.detach().float().cpu().numpy(), ro) < max(tf, tol_f), tag
        o.backward(go.cuda().to(o.dtype))
        e = c_oracle.bwd(q.float().numpy(), k.float().numpy(), v.float().numpy(), go.float().numpy(), mask=mask, p=p)
        for t, rr, n in zip((qq, kk, vv), e, ("dq", "dk", "dv")):
            assert rel_err(t.grad.float().cpu().numpy(), rr, atol=2e-2) < tb, tag + (n,)


# ---- 32x32x16-tile kernels (N >= 256): forward, dQ, dK/dV against the C oracle --------------------------------------
Q32_CASES = [
    # (B, H, Nq, Nk, D), p, mask
    ((1, 8, 256, 256, 64), 2, True), ((2, 3, 300, 300, 64), 2, True), ((1, 2, 333, 333, 128), 2, True),
    ((1, 5, 520, 520, 32), 1, True), ((1, 2, 384, 384, 80), 2, True), ((1, 16, 640, 640, 64), 2, False),
    ((1, 2, 257, 400, 64), 2, False), ((1, 3, 500, 290, 128), 1, False), ((1, 2, 1100, 1100, 64), 2, True),
    # head sizes above 128: the 16-row tiles with 256-column images (fp32 / fp16 gradients: vector-ALU tiles)
    ((1, 2, 300, 300, 256), 2, True), ((1, 2, 257, 400, 192), 2, False), ((1, 3, 520, 520, 160), 1, True),
]


@pytest.mark.parametrize("dt,tf,tb", [(torch.float32, TOL_FWD, TOL_BWD), (torch.bfloat16, 8e-3, 2.5e-2), (torch.float16, 2e-3, 5e-3)])
@pytest.mark.parametrize("shape,p,mask", Q32_CASES)
def test_wide_tile_kernels_forward_backward(shape, p, mask, dt, tf, tb):
    """quadratic family forced, sizes that route to fastmax_quad32_mfma.hip / fastmax_quad32_bwd.hip: ragged lengths,
    padded head sizes, head counts that are / are not a multiple of the 8 XCDs, N_q != N_k (unmasked)"""
    from attention_mechanisms.fastmax import fastmax
    from oracle import c_oracle
    B, H, Nq, Nk, D = shape
    g = torch.Generator().manual_seed(Nq + Nk + D + p)
    q, go = (torch.randn(B, H, Nq, D, generator=g).to(dt) for _ in range(2))
    k, v = (torch.randn(B, H, Nk, D, generator=g).to(dt) for _ in range(2))
    _force("quadratic_mfma")
    qq, kk, vv = (t.cuda().requires_grad_(True) for t in (q, k, v))
    o = fastmax(qq, kk, vv, mask=mask, p=p)
    ro, _ = c_oracle.fwd(q.float().numpy(), k.float().numpy(), v.float().numpy(), mask=mask, p=p)
    assert rel_err(o.detach().float().cpu().numpy(), ro) < (tf if o.dtype == dt else max(tf, TOL_FWD))
    o.backward(go.cuda().to(o.dtype))
    e = c_oracle.bwd(q.float().numpy(), k.float().numpy(), v.float().numpy(), go.float().numpy(), mask=mask, p=p)
    for t, rr, n in zip((qq, kk, vv), e, ("dq", "dk", "dv")):
        assert t.grad.dtype == dt
        assert rel_err(t.grad.float().cpu().numpy(), rr, atol=2e-2) < tb, n


def test_narrow_tile_kernels_still_serve_long_sequences():
    """FASTMAX_QUAD32=0 / FASTMAX_QUAD32_BWD=0 (read once per process) keep the 16-row-tile kernels on every size:
    run them at N >= 256 in a child process against the C oracle"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
from conftest import rel_err
from attention_mechanisms.fastmax import fastmax
from fastmax_experiments_amd import _lib, ops
from oracle import c_oracle
ops.set_forced_path(_lib.PATH_QUADRATIC_MFMA)
for (B, H, N, D, p, mask, dt, tf, tb) in [(1, 3, 300, 64, 2, True, torch.float32, 2e-4, 1e-3), (1, 2, 520, 128, 2, True, torch.bfloat16, 8e-3, 2.5e-2),
                                          (1, 8, 384, 32, 1, False, torch.float16, 2e-3, 5e-3)]:
    g = torch.Generator().manual_seed(N)
    q, k, v, go = (torch.randn(B, H, N, D, generator=g).to(dt) for _ in range(4))
    qq, kk, vv = (t.cuda().requires_grad_(True) for t in (q, k, v))
    o = fastmax(qq, kk, vv, mask=mask, p=p)
    ro, _ = c_oracle.fwd(q.float().numpy(), k.float().numpy(), v.float().numpy(), mask=mask, p=p)
    assert rel_err(o.detach().float().cpu().numpy(), ro) < max(tf, 2e-4), (N, "fwd")
    o.backward(go.cuda().to(o.dtype))
    e = c_oracle.bwd(q.float().numpy(), k.float().numpy(), v.float().numpy(), go.float().numpy(), mask=mask, p=p)
    for t, rr in zip((qq, kk, vv), e):
        assert rel_err(t.grad.float().cpu().numpy(), rr, atol=2e-2) < tb, (N, "bwd")
print("narrow ok")
''' % (root, root)
    env = dict(os.environ, FASTMAX_QUAD32="0", FASTMAX_QUAD32_BWD="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "narrow ok" in r.stdout, r.stdout + r.stderr


def _prologue_chain_fp64(x, gyn):
    """gradient wrt x of y = (x - mean_D x) / max_n ||x_n - mean_D x_n|| given dL/dy (float64 autograd over fastmax_hack.py:38-43)"""
    xr = torch.from_numpy(np.asarray(x, dtype=np.float64)).requires_grad_(True)
    xc = xr - xr.mean(-1, keepdim=True)
    y = xc / xc.norm(dim=-1).amax(-1)[..., None, None]
    y.backward(torch.from_numpy(np.asarray(gyn, dtype=np.float64)))
    return xr.grad.numpy()


@pytest.mark.parametrize("dt,tf,tb", [(torch.float32, 2 * TOL_FWD, 4 * TOL_BWD), (torch.bfloat16, 1.6e-2, 3e-2), (torch.float16, 3e-3, 8e-3)])
@pytest.mark.parametrize("shape", [(1, 2, 1024, 64), (2, 3, 700, 32), (1, 2, 2100, 128), (1, 40, 512, 64), (1, 2, 600, 48), (8, 48, 512, 32)])
def test_linearmax_training_route_with_the_prologue_inside_the_scans(shape, dt, tf, tb):
    """masked p=1 linearmax with gradients, N >= 512: ONE autograd node on the raw q, k, v (fastmax_hip_linearmax_forward_auto +
    fastmax_hip_linearmax_backward: the scans normalise while staging, no normalised copy is stored) against the C oracle's scan
    on float64-normalised inputs with the chain rule through the prologue, and against the two-node route (normalize_cast +
    fastmax) it replaces.  D = 128 exists for bf16 only (other dtypes keep the two-node route there: also checked).
    The forward's statistics carry the rows n* and BOTH scan kernels apply the prologue's backward to their own tiles (dq, dk
    leave as gradients wrt the raw tensors; one-row fix-ups add the dL/dM term) -- with the sequence split (few heads: statistics
    on the state pass) and without it (the last shape: paired statistics pass)."""
    import importlib
    fh = importlib.import_module("fastmax_experiments_amd.attention_mechanisms.fastmax_hack")      # the module, not the function
    from oracle import c_oracle, fastmax_oracle as orc
    g = torch.Generator().manual_seed(shape[2] + shape[3])
    q = (torch.randn(shape, generator=g) * 1.3 + 0.2).to(dt)
    k = (torch.randn(shape, generator=g) * 0.7).to(dt)
    v, go = (torch.randn(shape, generator=g).to(dt) for _ in range(2))
    res = {}
    for fused in (True, False):
        fh.FUSED_TRAINING = fused
        try:
            qq, kk, vv = (t.cuda().requires_grad_(True) for t in (q, k, v))
            o = fh.fastmax_hack(qq, kk, vv, p=1, mask=True)
            o.backward(go.cuda())
            res[fused] = [t.detach().float().cpu().numpy() for t in (o, qq.grad, kk.grad, vv.grad)]
            assert o.dtype == dt and qq.grad.dtype == dt
        finally:
            fh.FUSED_TRAINING = True
    qn, kn, vn, gn = (t.float().numpy() for t in (q, k, v, go))
    qq64, kk64 = orc.normalize_qk(qn, kn)
    ro, _ = c_oracle.fwd(qq64.astype(np.float32), kk64.astype(np.float32), vn, mask=True, nt=1.0, p=1)
    dqn, dkn, dv = c_oracle.bwd(qq64.astype(np.float32), kk64.astype(np.float32), vn, gn, mask=True, nt=1.0, p=1)
    want = [ro, _prologue_chain_fp64(qn, dqn), _prologue_chain_fp64(kn, dkn), dv]
    for fused in (True, False):
        for got, w, name, tol in zip(res[fused], want, ("o", "dq", "dk", "dv"), (tf, tb, tb, tb)):
            assert rel_err(got, w) < tol, (fused, name)


def test_linearmax_training_route_on_strided_storage():
    """q, k, v stored (B, N, H, D) and handed over as (B, H, N, D) views (what a fused QKV projection produces): the one-node route
    reads them through their strides -- same bits as on contiguous copies, forward and all three gradients"""
    from attention_mechanisms.fastmax_hack import fastmax_hack
    B, H, N, D = 2, 4, 1024, 64
    g = torch.Generator().manual_seed(21)
    base = [torch.randn(B, N, H, D, generator=g).to(torch.bfloat16).cuda() for _ in range(3)]
    go = torch.randn(B, H, N, D, generator=g).to(torch.bfloat16).cuda()
    res = []
    for strided in (True, False):
        leaves = [t.clone().requires_grad_(True) for t in base]
        q, k, v = (t.transpose(1, 2) if strided else t.transpose(1, 2).contiguous() for t in leaves)
        o = fastmax_hack(q, k, v, p=1, mask=True)
        o.backward(go)
        res.append([o.detach()] + [t.grad.clone() for t in leaves])
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_grouped_linearmax_training_route_without_normalised_copies():
    """grouped-query heads on the one-node route: q, v (B*G, rep, N, D) (v a stride-0 group view), K at its G heads -- same
    output and gradients as the two-node grouped route"""
    import importlib
    fh = importlib.import_module("fastmax_experiments_amd.attention_mechanisms.fastmax_hack")      # the module, not the function
    B, G, rep, N, D = 2, 2, 4, 640, 64
    g = torch.Generator().manual_seed(11)
    q0 = torch.randn(B * G, rep, N, D, generator=g).to(torch.bfloat16)
    k0 = (torch.randn(B, G, N, D, generator=g) * 1.5).to(torch.bfloat16)
    v0 = torch.randn(B, G, N, D, generator=g).to(torch.bfloat16)
    go = torch.randn(B * G, rep, N, D, generator=g).to(torch.bfloat16).cuda()
    res = {}
    for fused in (True, False):
        fh.FUSED_TRAINING = fused
        try:
            q, k, v = (t.cuda().requires_grad_(True) for t in (q0, k0, v0))
            vv = v.view(B * G, 1, N, D).expand(B * G, rep, N, D)
            o = fh.fastmax_hack_grouped(q, k, vv, rep, p=1)
            o.backward(go)
            res[fused] = [t.detach().float() for t in (o, q.grad, k.grad, v.grad)]
        finally:
            fh.FUSED_TRAINING = True
    for a, b, name in zip(res[True], res[False], ("o", "dq", "dk", "dv")):
        assert a.shape == b.shape and float((a - b).abs().max()) <= 2e-2 * float(b.abs().max()), name


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1.5e-2), (torch.float16, 2e-3)])
@pytest.mark.parametrize("shape", [(2, 3, 300, 64), (1, 2, 1000, 128), (1, 5, 17, 16), (1, 2, 260, 40), (2, 8, 513, 32), (1, 2, 300, 256),
                                   (1, 3, 130, 192)])
def test_linearmax_prologue_forward_backward_kernels(shape, dt, tol):
    """_NormalizeQK (fastmax_normalize.hip; D=40 in 16-bit takes the float32 kernel + tensor-op backward) against float64
    autograd over the reference's own formulation (fastmax_hack.py:38-43)"""
    from fastmax_experiments_amd.attention_mechanisms.fastmax_hack import _NormalizeQK
    g = torch.Generator().manual_seed(shape[2])
    x = torch.randn(shape, generator=g).to(dt)
    gy = torch.randn(shape, generator=g).to(dt)
    xr = x.double().requires_grad_(True)
    xc = xr - xr.mean(-1, keepdim=True)
    yr = xc / torch.linalg.norm(xc, dim=-1).max(dim=-1).values[..., None, None]
    yr.backward(gy.double())
    xx = x.cuda().requires_grad_(True)
    y = _NormalizeQK.apply(xx, 1)
    assert y.dtype == dt
    y.backward(gy.cuda())
    assert rel_err(y.detach().float().cpu().numpy(), yr.detach().numpy()) < tol
    assert xx.grad.dtype == dt
    assert rel_err(xx.grad.float().cpu().numpy(), xr.grad.numpy()) < tol


def test_linearmax_prologue_backward_is_reproducible():
    from fastmax_experiments_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 4, 2048, 64, generator=g).cuda()
    gy = torch.randn(2, 4, 2048, 64, generator=g).cuda()
    _, inv = ops.normalize_cast(x)
    a, b = ops.normalize_backward(x, gy, inv), ops.normalize_backward(x, gy, inv)
    assert torch.equal(a, b)


def test_forward_is_hip_graph_capturable():
    """launch-bound shapes (C2: five launches for 0.03 ms of work): after one eager warm-up call (one-time function attributes)
    the whole linearmax forward records into a HIP graph and replays with new inputs"""
    from attention_mechanisms.fastmax_hack import fastmax_hack
    g = torch.Generator().manual_seed(1)
    q, k, v = (torch.randn(16, 4, 1024, 32, generator=g).to(torch.bfloat16).cuda() for _ in range(3))
    with torch.no_grad():
        ref = fastmax_hack(q, k, v, p=1, mask=True)                # warm-up + eager result
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fastmax_hack(q, k, v, p=1, mask=True)
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = fastmax_hack(q, k, v, p=1, mask=True)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
        q2 = torch.randn(16, 4, 1024, 32, generator=g).to(torch.bfloat16).cuda()
        ref2 = fastmax_hack(q2, k, v, p=1, mask=True)
        q.copy_(q2)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref2)


def test_results_do_not_depend_on_stale_device_memory():
    """uninitialised-read hunt (tools/poison_check.py runs the full list): the allocator's free blocks are filled with
    zeros, then with NaN bit patterns, and the operators must return identical, finite results both times"""
    from attention_mechanisms.fastmax import fastmax
    from attention_mechanisms.fastmax_hack import fastmax_hack

    def poison(value):
        torch.cuda.synchronize()
        blocks = [torch.full((256 << 20,), value, dtype=torch.uint8, device="cuda") for _ in range(8)]
        del blocks
        torch.cuda.synchronize()

    def run(f, shape, dt, p):
        g = torch.Generator().manual_seed(shape[2])
        q, k, v, go = (torch.randn(shape, generator=g).to(dt).cuda() for _ in range(4))
        q.requires_grad_(True), k.requires_grad_(True), v.requires_grad_(True)
        o = f(q, k, v, p=p, mask=True)
        o.backward(go)
        return [t.float().cpu() for t in (o.detach(), q.grad, k.grad, v.grad)]

    for f, shape, dt, p in ((fastmax_hack, (16, 4, 1024, 32), torch.bfloat16, 1), (fastmax, (1, 32, 2048, 64), torch.float32, 1),
                            (fastmax, (2, 3, 777, 64), torch.bfloat16, 2), (fastmax_hack, (1, 2, 1100, 128), torch.bfloat16, 1)):
        poison(0)
        a = run(f, shape, dt, p)
        poison(0xFF)
        b = run(f, shape, dt, p)
        for x, y in zip(a, b):
            assert torch.isfinite(y).all() and torch.equal(x, y), (f.__name__, shape, dt, p)


@pytest.mark.parametrize("dt,tf,tb", [(torch.float32, TOL_FWD, TOL_BWD), (torch.bfloat16, 8e-3, 2.5e-2)])
@pytest.mark.parametrize("shape,p,mask", [((2, 3, 300, 50), 1, True), ((1, 2, 700, 33), 2, True), ((1, 4, 128, 5), 1, False),
                                          ((1, 2, 600, 100), 2, False), ((1, 2, 1030, 127), 1, True)])
def test_head_sizes_that_are_not_a_multiple_of_eight(shape, p, mask, dt, tf, tb):
    """zero-padded to the next multiple of 8 inside the autograd function (nt keeps the true D): matrix-core kernels
    instead of the vector-ALU fallbacks, same function"""
    from attention_mechanisms.fastmax import fastmax
    from fastmax_experiments_amd import _lib, ops
    from oracle import c_oracle
    g = torch.Generator().manual_seed(shape[2] + shape[3])
    q, k, v, go = (torch.randn(shape, generator=g).to(dt) for _ in range(4))
    qq, kk, vv = (t.cuda().requires_grad_(True) for t in (q, k, v))
    o = fastmax(qq, kk, vv, mask=mask, p=p)
    assert o.shape == tuple(shape)
    ro, _ = c_oracle.fwd(q.float().numpy(), k.float().numpy(), v.float().numpy(), mask=mask, p=p)
    assert rel_err(o.detach().float().cpu().numpy(), ro) < max(tf, TOL_FWD)
    o.backward(go.cuda().to(o.dtype))
    e = c_oracle.bwd(q.float().numpy(), k.float().numpy(), v.float().numpy(), go.float().numpy(), mask=mask, p=p)
    for t, rr, n in zip((qq, kk, vv), e, ("dq", "dk", "dv")):
        assert t.grad.shape == tuple(shape) and t.grad.dtype == dt
        assert rel_err(t.grad.float().cpu().numpy(), rr, atol=2e-2) < tb, n


def test_more_than_65535_heads_run_in_batch_slices():
    from attention_mechanisms.fastmax import fastmax
    from attention_mechanisms.fastmax_hack import fastmax_hack
    g = torch.Generator().manual_seed(2)
    q, k, v = (torch.randn(70000, 1, 16, 16, generator=g).cuda() for _ in range(3))
    o = fastmax(q, k, v, p=2)
    ref = torch.cat([fastmax(q[i:i + 30000], k[i:i + 30000], v[i:i + 30000], p=2) for i in range(0, 70000, 30000)])
    assert o.shape == q.shape and torch.equal(o, ref)
    assert fastmax_hack(q, k, v, p=1).shape == q.shape


@pytest.mark.parametrize("p,mask", [(1, True), (2, True), (2, False)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_strided_views_give_the_same_bits_as_contiguous_tensors(p, mask, dt):
    """(B, T, H, D) storage viewed as (B, H, T, D) (what a caller that skips the transpose copy hands over): the kernels take
    element strides for b, h, n -- forward and backward must match the contiguous run exactly"""
    from attention_mechanisms.fastmax import fastmax
    g = torch.Generator().manual_seed(7)
    B, T, H, D = 2, 640, 3, 64
    base = [torch.randn(B, T, H, D, generator=g).to(dt).cuda() for _ in range(3)]
    go = torch.randn(B, H, T, D, generator=g).to(dt).cuda()
    outs = []
    for contiguous in (False, True):
        q, k, v = (t.permute(0, 2, 1, 3) for t in base)
        if contiguous:
            q, k, v = (t.contiguous() for t in (q, k, v))
        q, k, v = (t.detach().requires_grad_(True) for t in (q, k, v))
        assert q.is_contiguous() == contiguous
        o = fastmax(q, k, v, mask=mask, p=p)
        o.backward(go.to(o.dtype))
        outs.append([o.detach(), q.grad, k.grad, v.grad])
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("shape,dt", [((1, 4, 2048, 64), torch.bfloat16), ((2, 2, 1536, 64), torch.float32),
                                      ((1, 3, 2048, 32), torch.float16), ((1, 2, 2048, 128), torch.bfloat16),
                                      ((8, 64, 512, 64), torch.bfloat16)])
def test_backward_reuses_the_forward_prefix_states(shape, dt):
    """fastmax_hip_backward_with_states: handing the forward's sequence-split states back gives bit-identical gradients to
    recomputing them; problems / layouts without a split report 0 state bytes and take the plain route"""
    from fastmax_experiments_amd import ops
    torch.manual_seed(5)
    q, k, v = (torch.randn(shape, device="cuda").to(dt) * 0.5 for _ in range(3))
    go = torch.randn(shape, device="cuda").to(dt)
    o, g, states = ops.forward(q, k, v, 1, True, 1.0, 0.0, dt, keep_states=True)
    many_heads = shape[0] * shape[1] >= 512
    assert (states is None) == many_heads                 # enough heads to fill the chip: no split, nothing to keep
    o2, g2 = ops.forward(q, k, v, 1, True, 1.0, 0.0, dt)
    assert torch.equal(o, o2) and torch.equal(g, g2)
    ref = ops.backward(q, k, v, o, g, go, 1, True, 1.0)
    got = ops.backward(q, k, v, o, g, go, 1, True, 1.0, states=states)
    for a, b in zip(got, ref):
        assert torch.equal(a, b)
    if states is not None:
        # a buffer that is too small, or states for a layout the forward would not have split, are ignored
        small = states[: states.numel() // 2]
        for a, b in zip(ops.backward(q, k, v, o, g, go, 1, True, 1.0, states=small), ref):
            assert torch.equal(a, b)
        # p = 2 has no carried state: nothing is kept
        assert ops.forward(q, k, v, 2, True, 1.0, 0.0, dt, keep_states=True)[2] is None


@pytest.mark.parametrize("shape,dt", [((2, 3, 700, 64), torch.bfloat16), ((1, 2, 256, 32), torch.float32), ((1, 1, 1030, 128), torch.float16)])
def test_normalize_cast_two_launch_form_matches_the_atomic_form(shape, dt):
    """fastmax_hip_normalize_cast with room for per-block maxima (two launches) gives the same bits as the zero + atomic-max +
    finish form it takes with the small workspace"""
    import ctypes
    from fastmax_experiments_amd import ops, _lib
    torch.manual_seed(11)
    x = (torch.randn(shape, device="cuda") * 3 + 0.5).to(dt)
    y, inv = ops.normalize_cast(x)
    L = _lib.lib()
    B, H, N, D = shape
    y2 = torch.empty_like(y)
    inv2 = torch.empty_like(inv)
    ws = torch.empty(L.fastmax_hip_normalize_workspace(B, H), dtype=torch.uint8, device="cuda")
    rc = L.fastmax_hip_normalize_cast(x.data_ptr(), ops._strides(x), ops._DT[dt], y2.data_ptr(), inv2.data_ptr(), B, H, N, D,
                                      ctypes.c_void_p(ws.data_ptr()), ws.numel(), ops._stream(x.device))
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(inv, inv2) and torch.equal(y, y2)
    ref = ops.normalize_stats(x)
    assert torch.equal(inv, ref)
