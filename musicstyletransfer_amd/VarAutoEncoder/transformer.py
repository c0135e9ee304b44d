"""TransformerConfig (reference VarAutoEncoder/transformer.py:8-21).

The Transformer blocks themselves (MultiHeadDotAttention :49-126, DualFeedForward :24-46, encoder /
decoder layers :129-201, positional_encodings :204-211) are not Python classes here: they are the kernel
sequence of musicstyletransfer_amd/engine.py (_layer_fwd / _layer_bwd) over the C-ABI."""
from typing import Optional

from .config import Config
from ..engine import positional_table as positional_encodings  # noqa: F401  (same arithmetic as :204-211)


class TransformerConfig(Config):
    def __init__(self, model_size: int, dropout: float, num_layers: int, num_heads: int, vocab_size: Optional[int] = None):
        super().__init__()
        self.model_size = model_size
        self.dropout = dropout
        self.num_layers = num_layers
        self.num_heads = num_heads
        self.vocab_size = vocab_size
