from .ocr_extractor import OCRExtractor

__all__ = ["OCRExtractor"]
