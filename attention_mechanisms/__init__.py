"""Top-level alias so the reference's import lines work unchanged with this repo on sys.path:

    from attention_mechanisms.fastmax import fastmax            # lit_gpt/model.py:24
    from attention_mechanisms.fastmax_hack import fastmax_hack  # lit_gpt/model.py:25
"""
