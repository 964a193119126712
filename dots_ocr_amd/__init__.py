"""dots.ocr on MI355X: a gfx950-native inference engine behind the reference's HF-style API."""
__version__ = "0.1.0"
