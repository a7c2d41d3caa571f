from .attentive_multispade import AttentiveMultiSpade  # noqa: F401
from .multispade import MultiSpade  # noqa: F401
from .spade import SPADE, AnySpadeResBlock  # noqa: F401
