"""MI355X-native fastmax / linearmax attention (hand-written HIP behind the reference's Python API).

    from fastmax_experiments_amd.attention_mechanisms.fastmax import fastmax
    from fastmax_experiments_amd.attention_mechanisms.fastmax_hack import fastmax_hack
"""
from .attention_mechanisms import fastattention_einops, fastmax, fastmax_hack  # noqa: F401

__all__ = ["fastmax", "fastmax_hack", "fastattention_einops"]
