"""Drop-in for the reference's ``ocrs`` package on the SLATE / Slot-Attention / IODINE paths:
``getattr(ocrs, config.ocr.name)(config.ocr, config.dataset)`` (train_ocr.py:37) and
``getattr(ocrs, name + "_Module")`` (utils/tools.py:327-331) resolve here."""
from .base import Base
from .iodine import Iodine, Iodine_Module
from .slate import SLATE, SLATE_Module

__all__ = ["Base", "SLATE", "SLATE_Module", "Iodine", "Iodine_Module"]
