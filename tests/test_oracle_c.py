"""Pin the plain-C oracle (oracle/fastmax_oracle.c) against the golden vectors from the
reference and against the numpy oracle.  CPU only."""
import numpy as np
import pytest

from conftest import golden_names, load_golden, rel_err
from oracle import c_oracle, fastmax_oracle as orc

NAMES = [n for n in golden_names("fm_") + golden_names("opt_") if "D128_p2" not in n]


def _nt(meta, D):
    kw = meta.get("kw", {})
    return orc.effective_normalize_term(D, kw.get("normalize_term", 8), kw.get("tensors_normalized", False))


@pytest.mark.parametrize("name", NAMES)
def test_c_oracle_vs_golden(name):
    d, meta = load_golden(name)
    nt = _nt(meta, d["q"].shape[-1])
    o, g = c_oracle.fwd(d["q"], d["k"], d["v"], mask=meta["mask"], nt=nt, p=meta["p"])
    assert rel_err(o, d["o"]) < 1e-11 and rel_err(g, d["g"]) < 1e-11
    dq, dk, dv = c_oracle.bwd(d["q"], d["k"], d["v"], d["grad_o"], mask=meta["mask"], nt=nt, p=meta["p"])
    for a, b in ((dq, d["dq"]), (dk, d["dk"]), (dv, d["dv"])):
        assert rel_err(a, b, atol=1e-3) < 1e-9


def test_c_oracle_d128_p2():
    d, meta = load_golden("fm_B1H2N33D128_p2_masked")
    o, g = c_oracle.fwd(d["q"], d["k"], d["v"], mask=True, p=2)
    assert rel_err(o, d["o"]) < 1e-11
    dq, dk, dv = c_oracle.bwd(d["q"], d["k"], d["v"], d["grad_o"], mask=True, p=2)
    assert max(rel_err(dq, d["dq"]), rel_err(dk, d["dk"]), rel_err(dv, d["dv"])) < 1e-9


@pytest.mark.parametrize("name", golden_names("decode_"))
def test_c_oracle_decode(name):
    d, meta = load_golden(name)
    o, _ = c_oracle.fwd(d["q"], d["k"], d["v"], mask=False, p=meta["p"])
    assert rel_err(o, d["o"]) < 1e-11


def test_c_oracle_linearmax_via_numpy_normalize():
    d, meta = load_golden("hack_masked_p1")
    y = c_oracle.normalize(d["q"])
    qn, _ = orc.normalize_qk(d["q"], d["k"])
    assert rel_err(y, qn) < 1e-6          # float32 output rounding only


def test_c_oracle_bad_p():
    q = np.zeros((1, 1, 2, 4), dtype=np.float32)
    with pytest.raises(ValueError):
        c_oracle.fwd(q, q, q, p=3)


def test_c_oracle_longer_sequence_matches_numpy():
    rng = np.random.default_rng(5)
    q, k, v, G = (rng.standard_normal((1, 2, 600, 32)).astype(np.float32) for _ in range(4))
    o, g = c_oracle.fwd(q, k, v, p=1)
    o2, g2 = orc.fastmax_fwd_factorized(q, k, v, p=1, chunk=64)
    assert rel_err(o, o2) < 1e-11 and rel_err(g, g2) < 1e-11
    dq, dk, dv = c_oracle.bwd(q, k, v, G, p=1)
    e = orc.fastmax_bwd_factorized(q, k, v, G, p=1, chunk=64)
    assert max(rel_err(dq, e[0]), rel_err(dk, e[1]), rel_err(dv, e[2])) < 1e-9
