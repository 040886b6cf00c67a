from fastmax_experiments_amd.attention_mechanisms.fastmax import fastattention_einops, fastmax  # noqa: F401
