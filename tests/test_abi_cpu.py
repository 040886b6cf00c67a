"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every
symbol include/fastmax_hip.h declares (no compute calls without a GPU), and the host logic
(argument validation, dtype rules, error behaviour) matches the reference's."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from fastmax_experiments_amd import _lib, build
    build.build()
    return _lib.lib()


def test_header_symbols_all_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "fastmax_hip.h")).read()
    declared = set(re.findall(r"\b(fastmax_hip_[a-z0-9_]+)\s*\(", hdr))
    from fastmax_experiments_amd import _lib
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s


def test_abi_version_and_error_strings(lib):
    assert lib.fastmax_hip_abi_version() == 8
    assert b"p should be 1 or 2" in lib.fastmax_hip_error_string(-1)
    assert lib.fastmax_hip_error_string(0) == b"ok"


def test_tuning_state_is_readable_and_wrong_result_variants_are_not_in_the_production_build(lib):
    """the bench line records the tune state; the timing-only kernel variants (matrix instructions / memory passes removed) exist
    only in -DFASTMAX_ABLATIONS builds"""
    assert lib.fastmax_hip_build_flags() == 0
    assert lib.fastmax_hip_tune_get(b"mfma_variant") == 200
    assert lib.fastmax_hip_tune_get(b"no_such_key") == -2 ** 31
    for v in (201, 204, 205, 206, 209, 119, 129, 219):
        assert lib.fastmax_hip_tune(b"mfma_variant", v) == -2
        assert lib.fastmax_hip_tune_get(b"mfma_variant") == 200
    assert lib.fastmax_hip_tune(b"mfma_variant", 121) == 0 and lib.fastmax_hip_tune_get(b"mfma_variant") == 121
    assert lib.fastmax_hip_tune(b"mfma_variant", 200) == 0


def test_problem_validation_without_gpu(lib):
    from fastmax_experiments_amd._lib import Problem, PATH_AUTO, PATH_QUADRATIC, PATH_RECURRENT, PATH_MFMA
    ok = Problem(2, 4, 256, 256, 64, 0, 0, 1, 1, 1 / 64, 1 / 8192, 256.0, PATH_AUTO)
    assert lib.fastmax_hip_select_path(ctypes.byref(ok)) in (PATH_RECURRENT, PATH_MFMA)
    p2 = Problem(2, 4, 256, 256, 64, 0, 0, 2, 1, 1 / 64, 1 / 8192, 256.0, PATH_AUTO)
    from fastmax_experiments_amd._lib import PATH_QUADRATIC_MFMA
    assert lib.fastmax_hip_select_path(ctypes.byref(p2)) == PATH_QUADRATIC_MFMA
    unm = Problem(2, 4, 1, 16, 64, 0, 0, 1, 0, 1 / 64, 1 / 8192, 1.0, PATH_AUTO)
    assert lib.fastmax_hip_select_path(ctypes.byref(unm)) == PATH_QUADRATIC           # N_q < 16: vector-ALU tiles
    odd = Problem(2, 4, 64, 64, 50, 0, 0, 2, 1, 1.0, 1.0, 0.0, PATH_AUTO)             # D % 4 != 0
    assert lib.fastmax_hip_select_path(ctypes.byref(odd)) == PATH_QUADRATIC
    bad_p = Problem(2, 4, 256, 256, 64, 0, 0, 3, 1, 1.0, 1.0, 0.0, PATH_AUTO)
    assert lib.fastmax_hip_select_path(ctypes.byref(bad_p)) == -1
    bad_shape = Problem(2, 4, 5, 16, 64, 0, 0, 1, 1, 1.0, 1.0, 0.0, PATH_AUTO)      # causal, Nq != Nk
    assert lib.fastmax_hip_select_path(ctypes.byref(bad_shape)) == -2
    big_d = Problem(2, 4, 16, 16, 256, 0, 0, 1, 1, 1.0, 1.0, 0.0, PATH_AUTO)                # head sizes up to 256: tile kernels only
    assert lib.fastmax_hip_select_path(ctypes.byref(big_d)) == PATH_QUADRATIC_MFMA
    big_d_scan = Problem(2, 4, 16, 16, 256, 0, 0, 1, 1, 1.0, 1.0, 0.0, PATH_RECURRENT)
    assert lib.fastmax_hip_select_path(ctypes.byref(big_d_scan)) == -2
    too_big_d = Problem(2, 4, 16, 16, 264, 0, 0, 1, 1, 1.0, 1.0, 0.0, PATH_AUTO)
    assert lib.fastmax_hip_select_path(ctypes.byref(too_big_d)) == -2
    # null pointers are rejected before anything is launched
    assert lib.fastmax_hip_forward(ctypes.byref(ok), None, None, None, None, None, None, None, None, None, 0, None) == -6
    # c w (B,H,Nq) floats, then the 32x32-tile kernels' w G copy (B,H,Nq,D) in the input dtype
    assert lib.fastmax_hip_backward_workspace(ctypes.byref(ok)) == 4 * 2 * 4 * 256 + 4 * 2 * 4 * 256 * 64


def test_reference_import_paths_and_signatures():
    import inspect
    from attention_mechanisms.fastmax import fastmax, fastattention_einops
    from attention_mechanisms.fastmax_hack import fastmax_hack
    sig = inspect.signature(fastmax)
    assert list(sig.parameters) == ["q", "k", "v", "mask", "normalize_term", "tensors_normalized", "p",
                                    "dropout_rate", "create_attn"]
    d = {k: v.default for k, v in sig.parameters.items() if v.default is not inspect.Parameter.empty}
    assert d == dict(mask=True, normalize_term=8, tensors_normalized=False, p=1, dropout_rate=0.0,
                     create_attn=False)
    hs = inspect.signature(fastmax_hack)
    assert list(hs.parameters) == ["q", "k", "v", "p", "mask"]
    assert hs.parameters["p"].default == 1 and hs.parameters["mask"].default is True
    assert issubclass(fastattention_einops, torch.autograd.Function)


def test_bad_p_raises_valueerror_like_reference():
    from attention_mechanisms.fastmax import fastmax
    q = torch.zeros(1, 1, 4, 8)
    with pytest.raises(ValueError):
        fastmax(q, q, q, p=3)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback():
    from attention_mechanisms.fastmax import fastmax
    q = torch.randn(1, 1, 4, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fastmax(q, q, q)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "fastmax_experiments_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("no oracle", ""), f"{f} mentions the oracle"
