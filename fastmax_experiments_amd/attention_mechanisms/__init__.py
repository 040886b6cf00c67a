"""Same module layout as the reference's ``attention_mechanisms`` package (imported at
lit_gpt/model.py:24-25): ``fastmax`` and ``fastmax_hack``."""
from .fastmax import fastattention_einops, fastmax  # noqa: F401
from .fastmax_hack import fastmax_hack  # noqa: F401
