"""Randomised parity cases (tools/fuzz_parity.py): random shapes, dtypes, orders, masks and memory layouts -- transposed
(B,N,H,D) storage, padded rows, grouped-query stride-0 head views -- forward and backward against the dense float64 oracle."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_random_shapes_and_layouts_against_the_oracle(monkeypatch):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_parity
    monkeypatch.setattr(sys, "argv", ["fuzz_parity.py", "30", "3"])
    assert fuzz_parity.main() == 0
