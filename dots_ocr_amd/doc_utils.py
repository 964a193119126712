"""PDF page rasterisation (reference dots_ocr/utils/doc_utils.py:20-60: every page rendered at 200 dpi, pages that would exceed
4500 px a side at 72 dpi instead).  Pure CPU, outside the accelerated path.

The reference renders with PyMuPDF (`fitz`).  When it is importable it is used, exactly as the reference does.  When it is not
(this image), a built-in rasteriser handles the documents an OCR pipeline is fed most: IMAGE-ONLY PDFs — scans — whose pages
paint one or more image XObjects (DCTDecode / CCITTFaxDecode / FlateDecode [+ PNG predictors] / raw; DeviceRGB, DeviceGray, DeviceCMYK, Indexed;
classic xref tables, object streams and compressed xrefs) placed with `cm` matrices; an invisible OCR text layer (render mode 3) is
ignored as it is invisible.  A page with visible text or vector painting is refused with PdfContentNotSupported instead of being
rendered wrongly.  Same names, arguments and dpi rules as the reference either way.
"""
from __future__ import annotations

import io
import math
import re
import zlib
from typing import Dict, List, Optional

from PIL import Image


MAX_STREAM_BYTES = 512 << 20        # decoded size cap of one stream (an A4 page at 600 dpi RGB is 104 MB)
MAX_PAGES = 100_000


class PdfContentNotSupported(NotImplementedError):
    """The built-in rasteriser only renders image-only (scanned) pages; install PyMuPDF for text / vector pages."""


def _have_fitz() -> bool:
    try:
        import fitz  # noqa: F401
        return True
    except Exception:
        return False


# ------------------------------------------------------------------------------------------------ reference path (PyMuPDF)
def fitz_doc_to_image(doc, target_dpi: int = 200, origin_dpi=None) -> Image.Image:
    """One page -> RGB PIL image at target_dpi; pages that would exceed 4500 px a side are rendered at 72 dpi instead
    (reference doc_utils.py:20-40).  `doc` is a PyMuPDF page, or a page of the built-in reader."""
    if isinstance(doc, _Page):
        return doc.render(target_dpi)
    import fitz
    pm = doc.get_pixmap(matrix=fitz.Matrix(target_dpi / 72, target_dpi / 72), alpha=False)
    if pm.width > 4500 or pm.height > 4500:
        pm = doc.get_pixmap(matrix=fitz.Matrix(1, 1), alpha=False)
    return Image.frombytes("RGB", (pm.width, pm.height), pm.samples)


def load_images_from_pdf(pdf_file, dpi: int = 200, start_page_id: int = 0, end_page_id=None) -> list:
    """reference doc_utils.py:43-60"""
    if _have_fitz():
        import fitz
        with fitz.open(pdf_file) as doc:
            pages = [doc[i] for i in range(doc.page_count)]
            return _select(pages, dpi, start_page_id, end_page_id)
    return _select(PdfDocument(pdf_file).pages, dpi, start_page_id, end_page_id)


def _select(pages, dpi, start_page_id, end_page_id):
    n = len(pages)
    end = end_page_id if end_page_id is not None and end_page_id >= 0 else n - 1
    if end > n - 1:
        print("end_page_id is out of range, use images length")
        end = n - 1
    return [fitz_doc_to_image(pages[i], target_dpi=dpi) for i in range(n) if start_page_id <= i <= end]


# ------------------------------------------------------------------------------------------------ built-in reader
class _Ref:
    __slots__ = ("num",)

    def __init__(self, num):
        self.num = num


class _Name(str):
    pass


class _Op(bytes):
    """an operator / keyword / closing delimiter (a literal string is plain bytes)"""


_WS = b" \t\r\n\x0c\x00"
_DELIM = b"()<>[]{}/%"


class _Lexer:
    """PDF object syntax (ISO 32000-1 §7.3): numbers, names, strings, arrays, dictionaries, references."""

    def __init__(self, data: bytes, pos: int = 0):
        self.d, self.p = data, pos

    def ws(self):
        d = self.d
        while self.p < len(d):
            c = d[self.p:self.p + 1]
            if c in (b" ", b"\t", b"\r", b"\n", b"\x0c", b"\x00"):
                self.p += 1
            elif c == b"%":
                while self.p < len(d) and d[self.p:self.p + 1] not in (b"\r", b"\n"):
                    self.p += 1
            else:
                break

    def token(self):
        """next object, or an operator / keyword as bytes; None at the end"""
        self.ws()
        d, p = self.d, self.p
        if p >= len(d):
            return None
        c = d[p:p + 1]
        if c == b"/":
            q = p + 1
            while q < len(d) and d[q] not in _WS and d[q] not in _DELIM:
                q += 1
            self.p = q
            raw = d[p + 1:q]
            raw = re.sub(rb"#([0-9A-Fa-f]{2})", lambda m: bytes([int(m.group(1), 16)]), raw)
            return _Name(raw.decode("latin-1"))
        if c == b"(":
            depth, q, out = 1, p + 1, bytearray()
            while q < len(d) and depth:
                ch = d[q]
                if ch == 0x5C:          # backslash
                    out += d[q:q + 2]
                    q += 2
                    continue
                depth += (ch == 0x28) - (ch == 0x29)
                if depth:
                    out.append(ch)
                q += 1
            self.p = q
            return bytes(out)
        if d[p:p + 2] == b"<<":
            self.p = p + 2
            out = {}
            while True:
                k = self.token()
                if k == b">>" or k is None:
                    return out
                out[k] = self.token()
        if c == b"<":
            q = d.index(b">", p)
            self.p = q + 1
            hx = re.sub(rb"\s", b"", d[p + 1:q])
            return bytes.fromhex((hx + b"0" * (len(hx) % 2)).decode())
        if c == b"[":
            self.p = p + 1
            out = []
            while True:
                v = self.token()
                if v == b"]" or v is None:
                    return out
                out.append(v)
        if d[p:p + 2] == b">>":
            self.p = p + 2
            return _Op(b">>")
        if c in (b"]", b"{", b"}", b")", b">"):
            self.p = p + 1
            return _Op(c)
        q = p
        while q < len(d) and d[q] not in _WS and d[q] not in _DELIM:
            q += 1
        self.p = max(q, p + 1)
        word = d[p:self.p]
        if re.fullmatch(rb"[+-]?\d+", word):
            # "N G R" reference?
            save = self.p
            m = re.match(rb"\s+(\d+)\s+R(?![A-Za-z0-9])", d[self.p:self.p + 32])
            if m and not word.startswith((b"+", b"-")):
                self.p += m.end()
                return _Ref(int(word))
            self.p = save
            return int(word)
        if re.fullmatch(rb"[+-]?(\d+\.\d*|\.\d+)", word):
            return float(word)
        if word == b"true":
            return True
        if word == b"false":
            return False
        if word == b"null":
            return None
        return _Op(word)                              # operator / keyword


def _png_unpredict(data: bytes, columns: int, colors: int, bpc: int) -> bytes:
    bpp = max(1, colors * bpc // 8)
    stride = (columns * colors * bpc + 7) // 8
    out, prev = bytearray(), bytearray(stride)
    for r in range(0, len(data) - stride, stride + 1):
        ft, row = data[r], bytearray(data[r + 1:r + 1 + stride])
        if ft == 1:
            for i in range(bpp, stride):
                row[i] = (row[i] + row[i - bpp]) & 255
        elif ft == 2:
            for i in range(stride):
                row[i] = (row[i] + prev[i]) & 255
        elif ft == 3:
            for i in range(stride):
                row[i] = (row[i] + (((row[i - bpp] if i >= bpp else 0) + prev[i]) >> 1)) & 255
        elif ft == 4:
            for i in range(stride):
                a = row[i - bpp] if i >= bpp else 0
                b, c = prev[i], (prev[i - bpp] if i >= bpp else 0)
                pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                row[i] = (row[i] + (a if pa <= pb and pa <= pc else (b if pb <= pc else c))) & 255
        out += row
        prev = row
    return bytes(out)


def _ccitt_image(data: bytes, parms: dict, width: int, height: int, get) -> Image.Image:
    """CCITTFaxDecode stream -> PIL image, by wrapping the fax data in a one-strip TIFF (Pillow's libtiff decodes Group 3 / 4).
    PDF semantics (ISO 32000-1 Table 11): the decoder writes 1 for a black run when BlackIs1, else 0; in DeviceGray a 1 bit is white.
    libtiff writes 1 for a black run, so Photometric = BlackIsZero reproduces BlackIs1 = true and WhiteIsZero the default."""
    import struct
    from PIL import features
    if not features.check("libtiff"):
        raise PdfContentNotSupported("CCITT fax images need Pillow with libtiff")
    k = int(get(parms.get("K", 0)) or 0)
    cols = int(get(parms.get("Columns", 1728)) or 1728)
    rows = int(get(parms.get("Rows", 0)) or 0) or height
    black_is_1 = bool(get(parms.get("BlackIs1", False)))
    if get(parms.get("EncodedByteAlign", False)) and k < 0:
        raise PdfContentNotSupported("byte-aligned Group 4 fax data")
    tags = [(256, 4, cols), (257, 4, rows), (258, 3, 1), (259, 3, 4 if k < 0 else 3), (262, 3, 1 if black_is_1 else 0), (266, 3, 1),
            (273, 4, 0), (277, 3, 1), (278, 4, rows), (279, 4, len(data))]
    if k >= 0:
        tags.append((292, 4, (1 if k > 0 else 0) | (4 if get(parms.get("EncodedByteAlign", False)) else 0)))
    tags.sort()
    ifd_len = 2 + 12 * len(tags) + 4
    strip_off = 8 + ifd_len
    out = bytearray(b"II*\x00" + struct.pack("<I", 8) + struct.pack("<H", len(tags)))
    for tag, typ, val in tags:
        if tag == 273:
            val = strip_off
        out += struct.pack("<HHI", tag, typ, 1) + (struct.pack("<HH", val, 0) if typ == 3 else struct.pack("<I", val))
    out += struct.pack("<I", 0) + data
    im = Image.open(io.BytesIO(bytes(out)))
    im.load()
    if im.size != (width, height):
        im = im.crop((0, 0, width, height))
    return im


class PdfDocument:
    """Just enough of a PDF reader to walk the page tree and fetch image XObjects."""

    def __init__(self, path_or_bytes):
        self.data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
        if not self.data.lstrip()[:5] == b"%PDF-":
            raise ValueError("not a PDF file")
        self.objs: Dict[int, object] = {}
        self.streams: Dict[int, bytes] = {}
        self._scan()
        root = self._root()
        self.pages: List[_Page] = []
        self._walk(self.get(root.get("Pages")), {})

    # ---- objects: a linear scan of "N G obj" (robust against broken xref tables; the last definition of a number wins)
    def _scan(self):
        d = self.data
        for m in re.finditer(rb"(?<![0-9])(\d+)\s+(\d+)\s+obj\b", d):
            lx = _Lexer(d, m.end())
            try:
                val = lx.token()
            except Exception:
                continue
            num = int(m.group(1))
            self.objs[num] = val
            lx.ws()
            if isinstance(val, dict) and d[lx.p:lx.p + 6] == b"stream":
                q = lx.p + 6
                q += 2 if d[q:q + 2] == b"\r\n" else 1
                length = self.get(val.get("Length"))
                end = q + length if isinstance(length, int) and d[q + length:q + length + 12].lstrip()[:9] == b"endstream" else d.index(b"endstream", q)
                self.streams[num] = d[q:end]
        for num, val in list(self.objs.items()):          # object streams (PDF 1.5)
            if isinstance(val, dict) and val.get("Type") == "ObjStm":
                body = self.stream(num)
                n, first = self.get(val.get("N")), self.get(val.get("First"))
                if not (isinstance(n, int) and isinstance(first, int) and 0 <= first <= len(body) and 0 <= n <= first):
                    raise ValueError("malformed PDF object stream header")        # every (number, offset) pair takes >= 1 byte of the header
                head = _Lexer(body[:first])
                pairs = [(head.token(), head.token()) for _ in range(n)]
                for onum, off in pairs:
                    self.objs.setdefault(onum, _Lexer(body, first + off).token())

    def _root(self) -> dict:
        for m in reversed(list(re.finditer(rb"trailer", self.data))):
            t = _Lexer(self.data, m.end()).token()
            if isinstance(t, dict) and "Root" in t:
                return self.get(t["Root"])
        for val in self.objs.values():                    # compressed xref: the XRef stream dictionary carries /Root
            if isinstance(val, dict) and val.get("Type") == "XRef" and "Root" in val:
                return self.get(val["Root"])
        for val in self.objs.values():
            if isinstance(val, dict) and val.get("Type") == "Catalog":
                return val
        raise ValueError("PDF has no document catalog")

    def get(self, v):
        seen = 0
        while isinstance(v, _Ref) and seen < 32:
            v, seen = self.objs.get(v.num), seen + 1
        return v

    def stream(self, num: int) -> bytes:
        """decoded stream bytes of object `num` (all filters but image codecs)"""
        dic, raw = self.objs[num], self.streams[num]
        filters, parms = self.get(dic.get("Filter")), self.get(dic.get("DecodeParms"))
        filters = [] if filters is None else (filters if isinstance(filters, list) else [filters])
        parms = parms if isinstance(parms, list) else [parms] * len(filters)
        for f, pr in zip(filters, parms):
            f, pr = self.get(f), self.get(pr) or {}
            if f in ("FlateDecode", "Fl"):
                dec = zlib.decompressobj()
                raw = dec.decompress(raw, MAX_STREAM_BYTES)          # bounded: a few KB of input must not become gigabytes
                if dec.unconsumed_tail:
                    raise ValueError(f"PDF stream inflates beyond {MAX_STREAM_BYTES >> 20} MiB")
                if self.get(pr.get("Predictor", 1)) >= 10:
                    raw = _png_unpredict(raw, self.get(pr.get("Columns", 1)), self.get(pr.get("Colors", 1)), self.get(pr.get("BitsPerComponent", 8)))
                elif self.get(pr.get("Predictor", 1)) == 2:
                    raise PdfContentNotSupported("TIFF predictor")
            elif f in ("ASCIIHexDecode", "AHx"):
                raw = bytes.fromhex(re.sub(rb"[^0-9A-Fa-f]", b"", raw.split(b">")[0]).decode())
            elif f in ("ASCII85Decode", "A85"):
                import base64
                raw = base64.a85decode(raw.strip().removesuffix(b"~>").removeprefix(b"<~") if hasattr(bytes, "removesuffix") else raw, adobe=False)
            elif f in ("DCTDecode", "DCT", "JPXDecode", "CCITTFaxDecode", "CCF"):
                break                                     # image codec: left to PIL (image())
            else:
                raise PdfContentNotSupported(f"stream filter {f}")
            if len(raw) > MAX_STREAM_BYTES:
                raise ValueError(f"PDF stream decodes beyond {MAX_STREAM_BYTES >> 20} MiB")
        return raw

    def _walk(self, node, inherited, _path=None, _depth=0):
        if not isinstance(node, dict):
            return
        # only the ANCESTORS on the current path can close a cycle: the same page object listed twice under /Kids (some generators do
        # that) is a repeated page, not a malformed tree; depth and the page count bound everything else
        _path = set() if _path is None else _path
        if id(node) in _path or _depth > 64 or len(self.pages) >= MAX_PAGES:      # /Kids cycles, absurd nesting, page bombs
            raise ValueError("malformed PDF page tree (cycle, depth > 64 or too many pages)")
        inh = dict(inherited)
        for k in ("MediaBox", "CropBox", "Resources", "Rotate"):
            if k in node:
                inh[k] = node[k]
        if node.get("Type") == "Pages" or "Kids" in node:
            _path.add(id(node))
            for kid in self.get(node.get("Kids")) or []:
                self._walk(self.get(kid), inh, _path, _depth + 1)
            _path.discard(id(node))
        else:
            self.pages.append(_Page(self, node, inh))

    @property
    def page_count(self) -> int:
        return len(self.pages)

    def __getitem__(self, i):
        return self.pages[i]

    # ---- image XObject -> PIL
    def image(self, ref) -> Image.Image:
        num = ref.num
        dic = self.objs[num]
        w, h = self.get(dic["Width"]), self.get(dic["Height"])
        filters = self.get(dic.get("Filter"))
        filters = [] if filters is None else (filters if isinstance(filters, list) else [filters])
        last = self.get(filters[-1]) if filters else None
        data = self.stream(num)
        if last in ("DCTDecode", "DCT", "JPXDecode"):
            im = Image.open(io.BytesIO(data))             # (Pillow undoes the inversion of Adobe CMYK JPEGs itself)
            im.load()
            return im.convert("RGB")
        if last in ("CCITTFaxDecode", "CCF"):             # bilevel scans: Group 3 / Group 4 fax coding
            parms = self.get(dic.get("DecodeParms"))
            parms = self.get(parms[-1]) if isinstance(parms, list) else parms
            im = _ccitt_image(data, parms or {}, w, h, self.get)
            dec = self.get(dic.get("Decode"))
            if isinstance(dec, list) and len(dec) == 2 and float(self.get(dec[0])) > float(self.get(dec[1])):
                from PIL import ImageChops
                im = ImageChops.invert(im.convert("L"))
            return im.convert("RGB")
        if self.get(dic.get("ImageMask")):
            raise PdfContentNotSupported("stencil image masks")
        bpc = self.get(dic.get("BitsPerComponent", 8))
        cs = self.get(dic.get("ColorSpace"))
        if isinstance(cs, list):
            kind = self.get(cs[0])
            if kind == "ICCBased":
                n = self.get(self.objs[cs[1].num].get("N")) if isinstance(cs[1], _Ref) else 3
                cs = {1: "DeviceGray", 3: "DeviceRGB", 4: "DeviceCMYK"}[n]
            elif kind == "Indexed":
                base, lookup = self.get(cs[1]), cs[3]
                if isinstance(base, list):
                    base = "DeviceRGB"
                pal = self.stream(lookup.num) if isinstance(lookup, _Ref) else self.get(lookup)
                if base != "DeviceRGB" or bpc != 8:
                    raise PdfContentNotSupported("indexed images other than 8-bit RGB palettes")
                im = Image.frombytes("P", (w, h), data)
                im.putpalette(pal[:768])
                return im.convert("RGB")
            elif kind in ("CalRGB", "CalGray"):
                cs = "DeviceRGB" if kind == "CalRGB" else "DeviceGray"
        if cs in ("DeviceGray", "G") and bpc == 1:
            return Image.frombytes("1", (w, h), data).convert("RGB")
        if bpc != 8:
            raise PdfContentNotSupported(f"{bpc}-bit image samples")
        mode = {"DeviceRGB": "RGB", "RGB": "RGB", "DeviceGray": "L", "G": "L", "DeviceCMYK": "CMYK", "CMYK": "CMYK"}.get(cs)
        if mode is None:
            raise PdfContentNotSupported(f"colour space {cs}")
        return Image.frombytes(mode, (w, h), data).convert("RGB")


_PAINT_OPS = {b"S", b"s", b"f", b"F", b"f*", b"B", b"B*", b"b", b"b*", b"sh"}


class _Page:
    def __init__(self, doc: PdfDocument, node: dict, inherited: dict):
        self.doc, self.node, self.inh = doc, node, inherited
        box = doc.get(inherited.get("CropBox") or inherited.get("MediaBox")) or [0, 0, 612, 792]
        x0, y0, x1, y1 = [float(doc.get(v)) for v in box]
        self.x0, self.y0 = min(x0, x1), min(y0, y1)
        self.width, self.height = abs(x1 - x0), abs(y1 - y0)
        self.rotate = int(doc.get(inherited.get("Rotate")) or 0) % 360

    def _content(self) -> bytes:
        c = self.node.get("Contents")
        refs = self.doc.get(c) if not isinstance(c, _Ref) else c
        refs = refs if isinstance(refs, list) else [c]
        return b"\n".join(self.doc.stream(r.num) for r in refs if isinstance(r, _Ref) and r.num in self.doc.streams)

    def placements(self):
        """[(image ref, ctm)] in paint order: a tiny content-stream interpreter (q / Q / cm / Do; invisible text ignored)."""
        res = self.doc.get(self.inh.get("Resources")) or {}
        xobjs = self.doc.get(res.get("XObject")) or {}
        ctm, stack, out, operands = (1.0, 0.0, 0.0, 1.0, 0.0, 0.0), [], [], []
        tr, in_text = 0, False
        lx = _Lexer(self._content())
        while True:
            t = lx.token()
            if t is None:
                break
            if not isinstance(t, _Op):
                operands.append(t)
                continue
            op = bytes(t)
            if op == b"q":
                stack.append(ctm)
            elif op == b"Q":
                ctm = stack.pop() if stack else ctm
            elif op == b"cm" and len(operands) >= 6:
                a, b, c, d, e, f = [float(v) for v in operands[-6:]]
                A, B, C, D, E, F = ctm
                ctm = (a * A + b * C, a * B + b * D, c * A + d * C, c * B + d * D, e * A + f * C + E, e * B + f * D + F)
            elif op == b"Do" and operands:
                ref = xobjs.get(operands[-1])
                dic = self.doc.get(ref)
                if isinstance(dic, dict) and dic.get("Subtype") == "Image":
                    out.append((ref, ctm))
                elif isinstance(dic, dict) and dic.get("Subtype") == "Form":
                    raise PdfContentNotSupported("form XObjects")
            elif op == b"BT":
                in_text = True
            elif op == b"ET":
                in_text = False
            elif op == b"Tr" and operands:
                tr = int(operands[-1])
            elif op in (b"Tj", b"TJ", b"'", b'"') and tr != 3:
                raise PdfContentNotSupported("visible text (not a scanned page): rendering it needs PyMuPDF")
            elif op in _PAINT_OPS:
                raise PdfContentNotSupported("vector painting (not a scanned page): rendering it needs PyMuPDF")
            elif op == b"BI":
                raise PdfContentNotSupported("inline images")
            operands = []
        return out

    def render(self, target_dpi: int = 200) -> Image.Image:
        def size(dpi):
            z = dpi / 72.0
            return max(1, math.ceil(self.width * z - 1e-3)), max(1, math.ceil(self.height * z - 1e-3)), z
        W, H, z = size(target_dpi)
        if W > 4500 or H > 4500:                          # reference doc_utils.py:33-36
            W, H, z = size(72)
        page = Image.new("RGB", (W, H), (255, 255, 255))
        for ref, (a, b, c, d, e, f) in self.placements():
            if abs(b) > 1e-6 * (abs(a) + 1) or abs(c) > 1e-6 * (abs(d) + 1):
                raise PdfContentNotSupported("rotated / skewed image placement")
            im = self.doc.image(ref)
            # the unit square maps to [e, e + a] x [f, f + d] in user space; image row 0 is the TOP (y = f + d)
            ux0, ux1 = sorted((e, e + a))
            uy0, uy1 = sorted((f, f + d))
            px0, px1 = (ux0 - self.x0) * z, (ux1 - self.x0) * z
            py0, py1 = (self.height - (uy1 - self.y0)) * z, (self.height - (uy0 - self.y0)) * z
            bw, bh = max(1, round(px1 - px0)), max(1, round(py1 - py0))
            if bw > 4 * W or bh > 4 * H:                       # a placement far larger than the page raster: refuse instead of allocating it
                raise ValueError(f"PDF image placement {bw}x{bh} px exceeds the page raster {W}x{H}")
            if a < 0:
                im = im.transpose(Image.FLIP_LEFT_RIGHT)
            if d < 0:
                im = im.transpose(Image.FLIP_TOP_BOTTOM)
            if im.size != (bw, bh):
                im = im.resize((bw, bh), Image.BICUBIC)
            page.paste(im, (round(px0), round(py0)))
        if self.rotate:
            page = page.rotate(-self.rotate, expand=True)
        return page
