"""Grounding-tokenizer input adapters (the reference's plugin contract, grounding_input/__init__.py):
a class with `prepare(batch) -> kwargs for PositionNet.forward`, `get_null_input()` and a `set` flag."""
