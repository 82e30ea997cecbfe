"""Drop-in for the reference's ``poolings`` package on the Transformer path:
``getattr(poolings, config.pooling.name)(ocr, config.pooling)`` and ``getattr(poolings, name + "_Module")`` (sb3s/ocr_extractor.py:20-35)."""
from .base import Base
from .transformer import Transformer, Transformer_Module

__all__ = ["Base", "Transformer", "Transformer_Module"]
