"""PDF page rasterisation (reference dots_ocr/utils/doc_utils.py:20-60).  Pure CPU, needs PyMuPDF,
outside the accelerated path: imported lazily, same names and dpi rules."""
from PIL import Image


def fitz_doc_to_image(doc, target_dpi: int = 200, origin_dpi=None) -> Image.Image:
    """One PyMuPDF page -> RGB PIL image at target_dpi; pages that would exceed 4500 px a side
    are rendered at 72 dpi instead."""
    import fitz
    pm = doc.get_pixmap(matrix=fitz.Matrix(target_dpi / 72, target_dpi / 72), alpha=False)
    if pm.width > 4500 or pm.height > 4500:
        pm = doc.get_pixmap(matrix=fitz.Matrix(1, 1), alpha=False)
    return Image.frombytes("RGB", (pm.width, pm.height), pm.samples)


def load_images_from_pdf(pdf_file, dpi: int = 200, start_page_id: int = 0, end_page_id=None) -> list:
    import fitz
    images = []
    with fitz.open(pdf_file) as doc:
        n = doc.page_count
        end = end_page_id if end_page_id is not None and end_page_id >= 0 else n - 1
        end = min(end, n - 1)
        for i in range(n):
            if start_page_id <= i <= end:
                images.append(fitz_doc_to_image(doc[i], target_dpi=dpi))
    return images
