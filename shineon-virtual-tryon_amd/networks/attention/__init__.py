from .sagan import SelfAttention

# registry mirrored from the reference (models/networks/attention/__init__.py:3)
ATTENTION_TYPES = {"sagan": SelfAttention}
