from fastmax_experiments_amd.attention_mechanisms.fastmax_hack import fastmax_hack  # noqa: F401
