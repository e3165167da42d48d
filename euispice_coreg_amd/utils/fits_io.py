"""
Minimal FITS image I/O for the drop-in `Alignment` API (the reference uses astropy.io.fits, which is not a
dependency here).  Reads / writes primary and IMAGE-extension HDUs (BITPIX 8/16/32/64/-32/-64, BSCALE/BZERO);
headers are plain insertion-ordered dicts.  Tile-compressed images (ZIMAGE binary tables) need astropy: when
astropy is importable it is used transparently, otherwise a clear error is raised.
"""
from __future__ import annotations

import mmap
import os

import numpy as np

BLOCK = 2880
_BITPIX_DTYPE = {8: ">u1", 16: ">i2", 32: ">i4", 64: ">i8", -32: ">f4", -64: ">f8"}


class Header(dict):
    """dict with FITS-ish conveniences (`copy()` returns a Header; comments are dropped)."""

    def copy(self):
        return Header(self)


def _parse_value(s: str):
    s = s.strip()
    if not s:
        return None
    if s[0] == "'":
        # a string: up to the closing quote ('' inside stands for one quote), trailing blanks insignificant
        end = s.find("'", 1)
        if end > 0 and (end + 1 >= len(s) or s[end + 1] != "'"):
            return s[1:end].rstrip()  # (the common case: no quote inside)
        end = 1
        out = []
        while end < len(s):
            if s[end] == "'":
                if end + 1 < len(s) and s[end + 1] == "'":
                    out.append("'")
                    end += 2
                    continue
                break
            out.append(s[end])
            end += 1
        return "".join(out).rstrip()
    k = s.find("/")
    if k >= 0:
        s = s[:k].strip()
    if s == "T":
        return True
    if s == "F":
        return False
    try:
        return int(s)
    except ValueError:
        pass
    try:
        return float(s)
    except ValueError:
        pass
    try:
        return float(s.replace("D", "E").replace("d", "e"))
    except ValueError:
        return s


def _read_header(f):
    hdr = Header()
    raw = b""
    last_key = None
    while True:
        block = f.read(BLOCK)
        if len(block) < BLOCK:
            if not raw and not block:
                return None, b""
            raise IOError("truncated FITS header")
        raw += block
        done = False
        text = block.decode("ascii", "replace")  # (one decode per block; cards are slices of it)
        for i in range(0, BLOCK, 80):
            key = text[i:i + 8].strip()
            if key == "END":
                done = True
                break
            if not key:
                continue
            if text[i + 8:i + 10] == "= " and key != "COMMENT" and key != "HISTORY":
                hdr[key] = _parse_value(text[i + 10:i + 80])
                last_key = key
                continue
            if key == "CONTINUE" and last_key is not None:
                # the long-string convention: a string value ending in '&' goes on in the next card's string
                prev = hdr.get(last_key)
                if isinstance(prev, str) and prev.endswith("&"):
                    part = _parse_value(text[i + 8:i + 80])
                    hdr[last_key] = prev[:-1] + (part if isinstance(part, str) else "")
                    continue
            last_key = None
        if done:
            break
    return hdr, raw


def _data_size(hdr):
    naxis = int(hdr.get("NAXIS", 0))
    if naxis == 0:
        return 0, ()
    shape = tuple(int(hdr["NAXIS%d" % (i + 1)]) for i in range(naxis))[::-1]
    n = int(np.prod(shape)) * abs(int(hdr["BITPIX"])) // 8
    n = n * int(hdr.get("GCOUNT", 1)) + int(hdr.get("PCOUNT", 0))
    return n, shape


def _decode(hdr, buf, nbytes, shape):
    """Pixels of an image HDU from its raw big-endian bytes (one pass: the byte swap), None for anything else."""
    is_image = hdr.get("SIMPLE") is not None or str(hdr.get("XTENSION", "")).strip() == "IMAGE"
    if not (is_image and shape):
        return None
    dt = _BITPIX_DTYPE[int(hdr["BITPIX"])]
    arr = np.frombuffer(buf, dtype=dt, count=nbytes // np.dtype(dt).itemsize).reshape(shape)  # (a view, no copy)
    bscale, bzero = hdr.get("BSCALE", 1), hdr.get("BZERO", 0)
    if bscale != 1 or bzero != 0:
        return arr.astype(np.float64) * bscale + bzero
    return arr.astype(arr.dtype.newbyteorder("="))


_GUNZIPPED = {}   # (abs path, mtime_ns, size) of a gzip-compressed FITS file -> its decompressed copy
_GUNZIP_DIR = None


def _plain_path(path):
    """`path`, or -- for a gzip-compressed file (`*.fits.gz`: astropy, hence the reference, opens them transparently) --
    the path of a decompressed copy made once per file state in a scratch directory that is removed at exit: everything
    downstream (memory-mapped data units, raw uploads, compressed-tile uploads) then works on it as on any file."""
    global _GUNZIP_DIR
    try:
        with open(path, "rb") as f:
            if f.read(2) != b"\x1f\x8b":
                return path
    except OSError:
        return path
    st = os.stat(path)
    key = (os.path.abspath(os.fspath(path)), st.st_mtime_ns, st.st_size)
    out = _GUNZIPPED.get(key)
    if out is None or not os.path.exists(out):
        import atexit
        import gzip
        import shutil
        import tempfile
        if _GUNZIP_DIR is None:
            base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
            _GUNZIP_DIR = tempfile.mkdtemp(prefix="coreg_gunzip_", dir=base)
            atexit.register(shutil.rmtree, _GUNZIP_DIR, True)
        fd, out = tempfile.mkstemp(suffix=".fits", dir=_GUNZIP_DIR)
        with os.fdopen(fd, "wb") as fo, gzip.open(path, "rb") as fi:
            shutil.copyfileobj(fi, fo, 1 << 22)
        _GUNZIPPED[key] = out
    return out


def _read_all(path, with_data=True, only=None):
    """[(header, data)] of every HDU.  with_data=False: headers only (data skipped with a seek); only=<window>: the pixel
    data of that HDU alone are read and decoded (an index, or an EXTNAME)."""
    hdus = []
    path = _plain_path(path)
    with open(path, "rb") as f:
        raw = []
        while True:
            hdr, _ = _read_header(f)
            if hdr is None:
                break
            nbytes, shape = _data_size(hdr)
            data = None
            padded = ((nbytes + BLOCK - 1) // BLOCK) * BLOCK
            if nbytes and (not with_data or only is not None):
                raw.append((f.tell(), nbytes, shape))
                f.seek(padded, os.SEEK_CUR)
            elif nbytes:
                raw.append(None)
                data = _decode(hdr, f.read(padded), nbytes, shape)
            else:
                raw.append(None)
            hdus.append((hdr, data))
        if with_data and only is not None and hdus:
            i = _select(hdus, only)
            if raw[i] is not None:
                pos, nbytes, shape = raw[i]
                f.seek(pos)
                hdus[i] = (hdus[i][0], _decode(hdus[i][0], f.read(nbytes), nbytes, shape))
    return hdus


def _select(hdus, window):
    if isinstance(window, str):
        for i, (h, _) in enumerate(hdus):
            if str(h.get("EXTNAME", "")).strip() == window:
                return i
        raise KeyError(f"no HDU with EXTNAME {window!r}")
    n = len(hdus)
    i = int(window)
    if i < 0:
        i += n
    if not 0 <= i < n:
        raise IndexError(f"HDU index {window} out of range ({n} HDUs)")
    return i


def _populate():
    """Flags for the mapping of a data unit that is about to be uploaded: NOT MAP_POPULATE -- the copy threads of the
    library take the page faults, in parallel, faster than mmap fills the page tables on one core (a file never mapped
    before: open + upload 1.68 -> 1.10 ms for 16 MiB, one mapped before 0.84 -> 0.77; profiles/fresh_file_upload.py).
    COREG_MMAP_POPULATE=1 fills them at mmap time."""
    return getattr(mmap, "MAP_POPULATE", 0) if os.environ.get("COREG_MMAP_POPULATE", "0") == "1" else 0


class RawImage:
    """The pixels of one image HDU exactly as the file stores them -- a read-only memory map of the data unit (big-endian
    elements) plus what decoding needs (BITPIX, BSCALE, BZERO).  The library uploads these bytes as they are and decodes
    them on the GPU (`coreg_set_small_fits`, `coreg_prepare_reference_*_fits`): no host pass over the pixels at all.
    `np.asarray(raw)` / `raw.decode()` give the pixels as `read_image` returns them, for host-side code."""

    def __init__(self, path, offset, nbytes, shape, hdr):
        self.path, self.shape, self.header = path, tuple(shape), hdr
        self.bitpix = int(hdr["BITPIX"])
        self.bscale, self.bzero = hdr.get("BSCALE", 1), hdr.get("BZERO", 0)
        self.nbytes = nbytes
        gran = mmap.ALLOCATIONGRANULARITY
        start = (offset // gran) * gran
        flags = mmap.MAP_SHARED | _populate()
        with open(path, "rb") as f:
            self._mm = mmap.mmap(f.fileno(), nbytes + (offset - start), flags=flags, prot=mmap.PROT_READ, offset=start)
        self._bytes = np.frombuffer(self._mm, dtype=np.uint8, count=nbytes, offset=offset - start)

    ndim = 2

    @property
    def ptr(self):
        return self._bytes.ctypes.data

    @property
    def dtype(self):
        """dtype of the decoded pixels (`decode()`)."""
        if self.bscale != 1 or self.bzero != 0:
            return np.dtype(np.float64)
        return np.dtype(_BITPIX_DTYPE[self.bitpix]).newbyteorder("=")

    def decode(self):
        return _decode(self.header, self._bytes, self.nbytes, self.shape)

    def __array__(self, dtype=None, copy=None):
        a = self.decode()
        return a if dtype is None else a.astype(dtype, copy=False)

    def close(self):
        self._bytes = None
        try:
            self._mm.close()
        except (BufferError, ValueError):
            pass  # a view is still alive somewhere: the map goes with it


def open_raw(path, window=-1):
    """RawImage of a 2-D image HDU of a local, uncompressed FITS file, with its header; None when that is not what
    `path` / `window` name (an in-memory pair, a URL, a tile-compressed or non-2-D HDU, an empty data unit): the caller
    then takes `read_image`."""
    if not isinstance(path, (str, os.PathLike)) or not os.path.isfile(path):
        return None
    try:
        path = _plain_path(path)
        hdus, spans = _scan(path)
        i = _select([(h, None) for h in hdus], window)
    except (IOError, KeyError, IndexError, ValueError, EOFError):
        return None
    hdr = hdus[i]
    is_image = hdr.get("SIMPLE") is not None or str(hdr.get("XTENSION", "")).strip() == "IMAGE"
    if spans[i] is None or not is_image or int(hdr.get("NAXIS", 0)) != 2 or int(hdr["BITPIX"]) not in _BITPIX_DTYPE:
        return None
    pos, nbytes, shape = spans[i]
    if int(hdr.get("GCOUNT", 1)) != 1 or int(hdr.get("PCOUNT", 0)) != 0 or nbytes == 0:
        return None
    if not (np.isfinite(float(hdr.get("BSCALE", 1))) and np.isfinite(float(hdr.get("BZERO", 0)))):
        return None
    return RawImage(os.fspath(path), pos, nbytes, shape, hdr)


# ---- tile-compressed images (the image lives in a binary table: what EUI level-1 / level-2 files hold) ----------------
_TFORM_BYTES = {"L": 1, "X": 1, "B": 1, "I": 2, "J": 4, "K": 8, "A": 1, "E": 4, "D": 8, "C": 8, "M": 16, "P": 8, "Q": 16}
# keywords of the table / of the compression convention that do not belong to the image header astropy presents
_TABLE_ONLY = ("XTENSION", "BITPIX", "NAXIS", "NAXIS1", "NAXIS2", "PCOUNT", "GCOUNT", "TFIELDS", "THEAP", "ZIMAGE",
               "ZTENSION", "ZBITPIX", "ZNAXIS", "ZPCOUNT", "ZGCOUNT", "ZCMPTYPE", "ZQUANTIZ", "ZDITHER0", "ZBLANK", "ZSCALE",
               "ZZERO", "ZSIMPLE", "ZEXTEND", "ZBLOCKED", "ZHECKSUM", "ZDATASUM", "CHECKSUM", "DATASUM")
_QUANTIZE = {"NO_DITHER": 1, "SUBTRACTIVE_DITHER_1": 2, "SUBTRACTIVE_DITHER_2": 3}


def _image_header_of_table(th):
    """The header of the IMAGE a compressed-image table stands for (as astropy's CompImageHDU.header presents it): the
    Z-keywords give BITPIX / NAXISn back, table and compression keywords go."""
    h = Header()
    h["BITPIX"] = int(th["ZBITPIX"])
    naxis = int(th["ZNAXIS"])
    h["NAXIS"] = naxis
    for i in range(naxis):
        h["NAXIS%d" % (i + 1)] = int(th["ZNAXIS%d" % (i + 1)])
    for k, v in th.items():
        if k in _TABLE_ONLY or k in h:
            continue
        if any(k.startswith(p) and k[len(p):].isdigit() for p in ("TTYPE", "TFORM", "TUNIT", "TDIM", "TNULL", "TSCAL",
                                                                  "TZERO", "TDISP", "ZNAXIS", "ZTILE", "ZNAME", "ZVAL")):
            continue
        h[k] = v
    return h


class CompressedImage:
    """One tile-compressed image HDU, NOT decompressed: the table's heap memory-mapped, the per-tile descriptors and
    scale / zero columns parsed, the compression parameters read.  The library uploads the COMPRESSED bytes and decodes
    them on the GPU (`coreg_set_small_tiled`, `coreg_prepare_reference_*_tiled`: cfitsio's RICE_1 codec and float
    dequantization restated, csrc/ricecomp.hpp).  `np.asarray(ci)` / `ci.decode()` decode on the host through the same
    code (plus zlib for tiles cfitsio stored gzipped, and for GZIP_1 / GZIP_2 images) -- no astropy involved."""

    ndim = 2

    def __init__(self, path, data_pos, table_hdr):
        th = table_hdr
        self.path, self.table_header = path, th
        self.header = _image_header_of_table(th)
        if int(th["ZNAXIS"]) != 2:
            raise NotImplementedError("tile-compressed images of dimension other than 2")
        self.zbitpix = int(th["ZBITPIX"])
        self.shape = (int(th["ZNAXIS2"]), int(th["ZNAXIS1"]))
        self.ztile = (int(th.get("ZTILE1", self.shape[1])), int(th.get("ZTILE2", 1)))
        self.cmptype = str(th["ZCMPTYPE"]).strip().upper()
        if self.cmptype == "RICE_ONE":  # (the name early cfitsio versions wrote; cfitsio reads both)
            self.cmptype = "RICE_1"
        params = {str(th["ZNAME%d" % i]).strip(): th["ZVAL%d" % i] for i in range(1, 20) if "ZNAME%d" % i in th}
        self.blocksize = int(params.get("BLOCKSIZE", 32))
        self.bytepix = int(params.get("BYTEPIX", 4))
        q = str(th.get("ZQUANTIZ", "NO_DITHER")).strip()
        if self.zbitpix < 0 and q == "NONE":
            self.quantize = 0  # lossless floats (GZIP only)
        elif self.zbitpix < 0:
            if q not in _QUANTIZE:
                raise NotImplementedError(f"ZQUANTIZ = {q!r}")
            self.quantize = _QUANTIZE[q]
        else:
            self.quantize = 0
        self.dither0 = int(th.get("ZDITHER0", 1))
        self.bscale, self.bzero = th.get("BSCALE", 1), th.get("BZERO", 0)
        # ---- the table: rows of fixed-size fields, variable-length arrays in the heap
        row_bytes, n_rows = int(th["NAXIS1"]), int(th["NAXIS2"])
        ntx = -(-self.shape[1] // self.ztile[0])
        nty = -(-self.shape[0] // self.ztile[1])
        if n_rows != ntx * nty:
            raise ValueError("compressed image: the table does not hold one row per tile")
        cols, off = {}, 0
        for i in range(1, int(th["TFIELDS"]) + 1):
            form = str(th["TFORM%d" % i]).strip()
            j = 0
            while j < len(form) and form[j].isdigit():
                j += 1
            rep, code = (int(form[:j]) if j else 1), form[j]
            if code not in _TFORM_BYTES:
                raise NotImplementedError(f"TFORM{i} = {form!r}")
            size = rep * _TFORM_BYTES[code] if code != "X" else (rep + 7) // 8
            cols[str(th["TTYPE%d" % i]).strip()] = (off, code, rep)
            off += size
        if off != row_bytes:
            raise ValueError("compressed image: TFORMn do not add up to NAXIS1")
        heap_off = int(th.get("THEAP", row_bytes * n_rows))
        heap_bytes = int(th.get("PCOUNT", 0)) - (heap_off - row_bytes * n_rows)
        total = heap_off + max(heap_bytes, 0)
        gran = mmap.ALLOCATIONGRANULARITY
        start = (data_pos // gran) * gran
        with open(path, "rb") as f:
            self._mm = mmap.mmap(f.fileno(), total + (data_pos - start), flags=mmap.MAP_SHARED | _populate(),
                                 prot=mmap.PROT_READ, offset=start)
        base = data_pos - start
        rows = np.frombuffer(self._mm, dtype=np.uint8, count=row_bytes * n_rows, offset=base).reshape(n_rows, row_bytes)
        self._heap = np.frombuffer(self._mm, dtype=np.uint8, count=max(heap_bytes, 0), offset=base + heap_off)
        self.n_tiles = n_rows

        def descriptors(name):
            if name not in cols:
                return None, None
            o, code, _ = cols[name]
            if code == "P":
                d = np.ascontiguousarray(rows[:, o:o + 8]).view(">i4").reshape(n_rows, 2)
            elif code == "Q":
                d = np.ascontiguousarray(rows[:, o:o + 16]).view(">i8").reshape(n_rows, 2)
            else:
                raise ValueError(f"column {name}: not a variable-length array")
            return d[:, 0].astype(np.int32), d[:, 1].astype(np.int64)

        def scalars(name, kw_default):
            if name in cols:
                o, code, _ = cols[name]
                dt = {"D": ">f8", "E": ">f4", "J": ">i4", "I": ">i2", "K": ">i8", "B": "u1"}[code]
                w = np.dtype(dt).itemsize
                return np.ascontiguousarray(rows[:, o:o + w]).view(dt).reshape(n_rows).astype(np.float64 if code in "DE" else np.int64)
            return kw_default

        self.tile_nbytes, self.tile_offset = descriptors("COMPRESSED_DATA")
        if self.tile_nbytes is None:
            raise ValueError("compressed image: no COMPRESSED_DATA column")
        self.gzip_nbytes, self.gzip_offset = descriptors("GZIP_COMPRESSED_DATA")
        if "UNCOMPRESSED_DATA" in cols and descriptors("UNCOMPRESSED_DATA")[0].any():
            raise NotImplementedError("compressed image with tiles in UNCOMPRESSED_DATA")
        self.zscale = scalars("ZSCALE", None)
        self.zzero = scalars("ZZERO", None)
        self.zscale0, self.zzero0 = float(th.get("ZSCALE", 1.0)), float(th.get("ZZERO", 0.0))
        if self.zbitpix < 0 and self.zscale is None and "ZSCALE" not in th:
            self.quantize = 0  # no scale anywhere: the tiles hold the floating-point values themselves (lossless GZIP)
        blank = scalars("ZBLANK", None)
        if blank is not None:
            if len(np.unique(blank)) > 1:
                raise NotImplementedError("compressed image with a per-tile ZBLANK column")
            self.has_blank, self.blank = True, int(blank[0])
        elif "ZBLANK" in th:
            self.has_blank, self.blank = True, int(th["ZBLANK"])
        elif "BLANK" in th and self.zbitpix > 0:
            self.has_blank, self.blank = True, int(th["BLANK"])
        else:
            self.has_blank, self.blank = False, 0
        if self.zbitpix > 0 and self.bscale == 1 and self.bzero == 0:
            #  astropy -- hence the reference -- turns BLANK into NaN only where BSCALE / BZERO make the image
            # floating point; unscaled integers keep the stored value, compressed or not (`_decode` of a plain image)
            self.has_blank = False

    # ---- what the library needs
    @property
    def on_gpu(self):
        """Decodable on the GPU as it is: RICE_1, every tile Rice-coded, float images quantized."""
        return (self.cmptype == "RICE_1" and bool((self.tile_nbytes > 0).all()) and self.bytepix in (1, 2, 4)
                and (self.zbitpix > 0 or self.quantize > 0) and self.zbitpix in (8, 16, 32, -32, -64))

    @property
    def dtype(self):
        if self.zbitpix == -32:
            return np.dtype(np.float32)
        if self.zbitpix == -64 or self.bscale != 1 or self.bzero != 0:
            return np.dtype(np.float64)
        return np.dtype({8: np.uint8, 16: np.int16, 32: np.int32}[self.zbitpix])

    def decode(self):
        """The pixels, as `read_image` returns them for a plain image of the same BITPIX (integers with BSCALE / BZERO:
        float64(stored) * BSCALE + BZERO)."""
        from .. import _lib
        ny, nx = self.shape
        if self.cmptype == "RICE_1":
            out = np.empty(self.shape, dtype=np.float32 if self.zbitpix == -32 else np.float64)
            status = _lib.decode_tiled_host(self, out)
            if (status == 1).any():
                raise IOError(f"{self.path}: corrupt Rice stream in tile(s) {np.flatnonzero(status == 1)[:5].tolist()}")
            for n in np.flatnonzero(status == 2):  # tiles cfitsio could not quantize: gzipped floats in the other column
                self._place(out, n, self._gunzip_tile(n, self.gzip_offset, self.gzip_nbytes, shuffled=False))
        elif self.cmptype in ("GZIP_1", "GZIP_2"):
            out = np.empty(self.shape, dtype=np.float32 if self.zbitpix == -32 else np.float64)
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(12, os.cpu_count() or 1)) as pool:  # (zlib releases the GIL)
                tiles = list(pool.map(lambda n: self._gunzip_tile(n, self.tile_offset, self.tile_nbytes,
                                                                  shuffled=self.cmptype == "GZIP_2"), range(self.n_tiles)))
            for n, t in enumerate(tiles):
                self._place(out, n, t)
        else:
            raise NotImplementedError(f"ZCMPTYPE = {self.cmptype!r} (RICE_1, GZIP_1 and GZIP_2 are read)")
        if self.zbitpix > 0 and self.bscale == 1 and self.bzero == 0:
            return out.astype(self.dtype)
        return out

    def _tile_box(self, n):
        ntx = -(-self.shape[1] // self.ztile[0])
        ty, tx = divmod(n, ntx)
        x0, y0 = tx * self.ztile[0], ty * self.ztile[1]
        return x0, y0, min(self.ztile[0], self.shape[1] - x0), min(self.ztile[1], self.shape[0] - y0)

    def _gunzip_tile(self, n, offsets, lengths, shuffled):
        import zlib
        x0, y0, tw, th = self._tile_box(n)
        raw = zlib.decompress(bytes(self._heap[int(offsets[n]):int(offsets[n]) + int(lengths[n])]), 31)
        if self.zbitpix < 0 and self.quantize > 0 and lengths is self.tile_nbytes:
            # (a QUANTIZED float image whose integers are gzipped instead of Rice-coded)
            raise NotImplementedError("GZIP-compressed quantized float images")
        dt = np.dtype(_BITPIX_DTYPE[self.zbitpix])
        if shuffled:
            raw = np.frombuffer(raw, dtype=np.uint8).reshape(dt.itemsize, tw * th).T.tobytes()
        a = np.frombuffer(raw, dtype=dt, count=tw * th).reshape(th, tw)
        if self.zbitpix > 0 and (self.bscale != 1 or self.bzero != 0):
            return a.astype(np.float64) * self.bscale + self.bzero
        return a

    def _place(self, out, n, tile):
        x0, y0, tw, th = self._tile_box(n)
        out[y0:y0 + th, x0:x0 + tw] = tile

    def __array__(self, dtype=None, copy=None):
        a = self.decode()
        return a if dtype is None else a.astype(dtype, copy=False)

    def close(self):
        self._heap = None
        try:
            self._mm.close()
        except (BufferError, ValueError):
            pass


def open_compressed(path, window=-1):
    """CompressedImage of a tile-compressed image HDU of a local file, or None when `path` / `window` do not name one."""
    if not isinstance(path, (str, os.PathLike)) or not os.path.isfile(path):
        return None
    try:
        path = _plain_path(path)
        hdus, spans = _scan(path)
        i = _select([(h, None) for h in hdus], window)
    except (IOError, KeyError, IndexError, ValueError, EOFError):
        return None
    hdr = hdus[i]
    if not (str(hdr.get("XTENSION", "")).strip() == "BINTABLE" and hdr.get("ZIMAGE") is True and spans[i] is not None):
        return None
    return CompressedImage(os.fspath(path), spans[i][0], hdr)


def open_cube(path, window=-1):
    """(array, header) of an image HDU of any dimension WITHOUT reading or decoding it: a read-only big-endian view of
    the memory-mapped data unit (NumPy converts what is actually touched, plane by plane, when it is used).  For the
    4-D SPICE windows of which a sweep needs a few wavelength planes.  Falls back to `read_image` (everything decoded)
    for anything that is not a plain, unscaled image HDU of a local file."""
    if isinstance(path, (str, os.PathLike)) and os.path.isfile(path):
        try:
            path = _plain_path(path)
            hdus, spans = _scan(path)
            i = _select([(h, None) for h in hdus], window)
            hdr = hdus[i]
            is_image = hdr.get("SIMPLE") is not None or str(hdr.get("XTENSION", "")).strip() == "IMAGE"
            plain = (spans[i] is not None and is_image and int(hdr["BITPIX"]) in _BITPIX_DTYPE
                     and hdr.get("BSCALE", 1) == 1 and hdr.get("BZERO", 0) == 0
                     and int(hdr.get("GCOUNT", 1)) == 1 and int(hdr.get("PCOUNT", 0)) == 0)
        except (IOError, KeyError, IndexError, ValueError):
            plain = False
        if plain:
            pos, nbytes, shape = spans[i]
            # (mapped with its page tables filled -- MAP_POPULATE -- so that the threads that sum the planes do not
            # take one page fault per 4 KiB each)
            gran = mmap.ALLOCATIONGRANULARITY
            start = (pos // gran) * gran
            with open(path, "rb") as f:
                mm = mmap.mmap(f.fileno(), nbytes + (pos - start),
                               flags=mmap.MAP_SHARED | (0 if os.environ.get("COREG_MMAP_POPULATE_CUBE", "1") == "0" else getattr(mmap, "MAP_POPULATE", 0)),
                               prot=mmap.PROT_READ, offset=start)
            dt = np.dtype(_BITPIX_DTYPE[int(hdr["BITPIX"])])
            arr = np.frombuffer(mm, dtype=dt, count=nbytes // dt.itemsize, offset=pos - start).reshape(tuple(shape))
            return arr, hdr
    return read_image(path, window)


def load_for_upload(path, window=-1):
    """(pixels, header) of one HDU for the library: a RawImage (nothing decoded, nothing read yet beyond the header)
    when the HDU is a plain image of a local file, else what `read_image` returns."""
    raw = open_raw(path, window)
    if raw is not None:
        return raw, raw.header
    comp = open_compressed(path, window)
    if comp is not None:
        return comp, comp.header
    return read_image(path, window)


_SCAN_CACHE = {}  # (abs path, mtime_ns, size) -> ([header], [span]) of the last few files looked at


def _scan(path, with_raw=False):
    """Headers of every HDU and where each data unit lies: ([header], [(offset, nbytes, shape) or None]).  A drop-in call
    asks for the same file's headers several times (header of the reference, header of the image, the raw view of
    either, the header blocks `write_corrected_fits` patches): parsed once per file state (path, modification time, size);
    the headers handed out are copies.  with_raw: also the header blocks as stored and where each HDU starts."""
    path = _plain_path(path)
    st = os.stat(path)
    key = (os.path.abspath(os.fspath(path)), st.st_mtime_ns, st.st_size)
    hit = _SCAN_CACHE.get(key)
    if hit is None:
        hdus, spans, raws, starts = [], [], [], []
        with open(path, "rb") as f:
            while True:
                starts.append(f.tell())
                hdr, raw = _read_header(f)
                if hdr is None:
                    starts.pop()
                    break
                nbytes, shape = _data_size(hdr)
                spans.append((f.tell(), nbytes, shape) if nbytes else None)
                f.seek(((nbytes + BLOCK - 1) // BLOCK) * BLOCK, os.SEEK_CUR)
                hdus.append(hdr)
                raws.append(raw)
        if len(_SCAN_CACHE) >= 16:  # (threads of a jitter session share it: every step tolerates the other's)
            try:
                _SCAN_CACHE.pop(next(iter(_SCAN_CACHE)), None)
            except (StopIteration, RuntimeError):
                pass
        hit = _SCAN_CACHE[key] = (hdus, spans, raws, starts)
    if with_raw:
        return [h.copy() for h in hit[0]], list(hit[1]), list(hit[2]), list(hit[3])
    return [h.copy() for h in hit[0]], list(hit[1])


def native_pixels(a):
    """Pixels as the sweep takes them: float32 data (BITPIX=-32, either byte order) as native float32, everything else as
    float64 (the reference's own cast, alignment.py:191 / :198 / :301 / :314; exact for float32)."""
    if isinstance(a, (RawImage, CompressedImage)):
        return a  # uploaded as stored, decoded on the GPU
    a = np.asarray(a)
    if a.dtype.kind == "f" and a.dtype.itemsize == 4:
        return np.ascontiguousarray(a, dtype=np.float32)
    return np.array(a, dtype=np.float64)


def _astropy_fits():
    try:
        import astropy.io.fits as afits  # optional: remote URLs, codecs this module does not read
        return afits
    except ImportError:
        return None


def read_image(path, window=-1):
    """(data, header) of one HDU.  `path` may also be a (data, header) pair already in memory.  Local files are read by
    this module -- plain images and tile-compressed ones (RICE_1, GZIP_1, GZIP_2: `CompressedImage`, pinned bit for bit
    against astropy / cfitsio) -- astropy, when installed, serves what is left (URLs, other codecs)."""
    if isinstance(path, (tuple, list)) and len(path) == 2:
        return np.asarray(path[0]), Header(path[1])
    afits = _astropy_fits()
    if os.path.exists(str(path)):
        try:
            comp = open_compressed(path, window)
            if comp is not None:
                return comp.decode(), comp.header
            hdus = _read_all(path, only=window)
            hdr, data = hdus[_select(hdus, window)]
            if data is not None:
                return data, hdr
        except NotImplementedError:
            if afits is None:
                raise
            hdr, data = None, None
        if afits is None:
            if "ZIMAGE" in hdr or str(hdr.get("XTENSION", "")).strip() == "BINTABLE":
                raise NotImplementedError("this table HDU is not an image this reader knows (astropy.io.fits is not installed)")
            raise ValueError(f"HDU {window!r} of {path} holds no image")
    elif afits is None:
        raise FileNotFoundError(path)
    with afits.open(path) as hl:
        hdu = hl[window]
        return np.array(hdu.data), Header({k: hdu.header[k] for k in hdu.header.keys() if k})


def read_header(path, window=-1):
    """Header of one HDU without decoding any pixel data (a tile-compressed image: the header of the image, as astropy
    presents it, not of the table that holds it)."""
    if isinstance(path, (tuple, list)) and len(path) == 2:
        return Header(path[1])
    if os.path.exists(str(path)):
        hdus = [(h, None) for h in _scan(path)[0]]
        hdr = hdus[_select(hdus, window)][0]
        if str(hdr.get("XTENSION", "")).strip() == "BINTABLE" and hdr.get("ZIMAGE") is True:
            return _image_header_of_table(hdr)
        return hdr
    afits = _astropy_fits()
    if afits is None:
        raise FileNotFoundError(path)
    with afits.open(path) as hl:
        hdu = hl[window]
        return Header({k: hdu.header[k] for k in hdu.header.keys() if k})


def file_identity(path, window=-1):
    """Hashable identity of (file contents as far as the OS tells, HDU): None for in-memory (data, header) pairs."""
    if not isinstance(path, (str, os.PathLike)):
        return None
    st = os.stat(path)
    return (os.path.abspath(os.fspath(path)), st.st_mtime_ns, st.st_size, window)


def _card(key, value):
    if isinstance(value, bool):
        v = "T" if value else "F"
        body = f"{v:>20}"
    elif isinstance(value, (int, np.integer)):
        body = f"{int(value):>20d}"
    elif isinstance(value, (float, np.floating)):
        r = repr(float(value)).upper()
        if "E" in r and "." not in r.split("E")[0]:
            r = r.replace("E", ".0E")
        body = f"{r:>20}"
    elif value is None:
        body = ""
    else:
        sv = str(value).replace("'", "''")
        if len(sv) > 68:
            # the long-string convention (CONTINUE cards): chunks of at most 67 characters + '&', cut so that a doubled
            # quote is never split
            chunks = []
            while len(sv) > 68:
                cut = 67
                if (cut - len(sv[:cut].rstrip("'"))) % 2 == 1:  # (quotes are doubled: an odd run would split a pair)
                    cut -= 1
                chunks.append(sv[:cut] + "&")
                sv = sv[cut:]
            chunks.append(sv)
            cards = [f"{key:<8}= '{chunks[0]}'".ljust(80)]
            cards += [f"CONTINUE  '{c}'".ljust(80) for c in chunks[1:]]
            return "".join(cards)
        body = "'" + f"{sv:<8}" + "'"
    return f"{key:<8}= {body}"[:80].ljust(80)


def _header_blob(i, bitpix, shape, hdr):
    """The header of HDU number i (0 = primary) as 2880-byte blocks: structural cards from (bitpix, shape), then the
    header's own cards in order (structural ones, NAXISn and over-long keys skipped)."""
    structural = ("SIMPLE", "XTENSION", "BITPIX", "NAXIS", "EXTEND", "PCOUNT", "GCOUNT", "BSCALE", "BZERO")
    cards = [_card("SIMPLE", True) if i == 0 else _card("XTENSION", "IMAGE"), _card("BITPIX", bitpix),
             _card("NAXIS", len(shape))]
    for k, n in enumerate(shape[::-1]):
        cards.append(_card("NAXIS%d" % (k + 1), int(n)))
    if i == 0:
        cards.append(_card("EXTEND", True))
    else:
        cards.append(_card("PCOUNT", 0))
        cards.append(_card("GCOUNT", 1))
    for k, v in hdr.items():
        if k in structural or k.startswith("NAXIS") or len(k) > 8:
            continue
        cards.append(_card(k, v))
    cards.append("END".ljust(80))
    blob = "".join(cards).encode("ascii")
    return blob + b" " * ((-len(blob)) % BLOCK)


def write_images(path, hdus, overwrite=True):
    """hdus: list of (data or None, header dict).  The first is written as the primary HDU."""
    if os.path.exists(path) and not overwrite:
        raise OSError(f"{path} exists")
    with open(path, "wb") as f:
        for i, (data, hdr) in enumerate(hdus):
            if data is None:
                bitpix, shape, raw = 8, (), b""
            else:
                data = np.asarray(data)
                bitpix = {"u1": 8, "i2": 16, "i4": 32, "i8": 64, "f4": -32, "f8": -64}[data.dtype.str[1:]]
                shape = data.shape
                raw = data.astype(_BITPIX_DTYPE[bitpix]).tobytes()
            f.write(_header_blob(i, bitpix, shape, hdr))
            if raw:
                f.write(raw + b"\0" * ((-len(raw)) % BLOCK))


_QUANTIZE_NAME = {v: k for k, v in _QUANTIZE.items()}


def noise_sigma(data):
    """Robust estimate of the pixel noise (cfitsio's third-order-difference estimator, in one pass over the whole image):
    1.4826 * median|2 a[i] - a[i-2] - a[i+2]| / sqrt(6) over rows, NaN-free differences only."""
    a = np.asarray(data, dtype=np.float64)
    if a.shape[1] < 5:
        return float(np.nanstd(a))
    d = np.abs(2.0 * a[:, 2:-2] - a[:, :-4] - a[:, 4:])
    d = d[np.isfinite(d)]
    return float(1.4826 * np.median(d) / np.sqrt(6.0)) if d.size else 0.0


def write_compressed_image(path, data, header, tile=None, quantize="SUBTRACTIVE_DITHER_1", quantize_level=16.0, scale=None,
                           dither0=1, blocksize=32, primary_header=None, overwrite=True):
    """Write `data` as ONE tile-compressed image HDU (RICE_1) after an empty primary HDU: the layout EUI level-1 /
    level-2 files have and astropy's CompImageHDU writes (the reference's writer, utils/Util.py:137-138) -- without astropy.
    Integer images (uint8, int16, uint16 via BZERO = 32768, int32) are lossless; floating-point images are quantized per
    tile, ZSCALE = `scale` or noise_sigma(data) / quantize_level, with `quantize` in NO_DITHER, SUBTRACTIVE_DITHER_1 / _2.
    tile: (ZTILE1, ZTILE2), default one image row per tile."""
    from .. import _lib
    if os.path.exists(path) and not overwrite:
        raise OSError(f"{path} exists")
    a = np.asarray(data)
    if a.ndim != 2:
        raise ValueError("image must be 2-D")
    ny, nx = a.shape
    tile = (nx, 1) if tile is None else (int(tile[0]), int(tile[1]))
    extra = {}
    if a.dtype.kind == "f":
        zbitpix, bytepix = (-32 if a.dtype.itemsize == 4 else -64), 4
        qcode = _QUANTIZE[quantize]
        if scale is None:
            sigma = noise_sigma(a)
            if not sigma > 0:
                raise ValueError("cannot estimate the noise of this image: pass scale=")
            scale = sigma / float(quantize_level)
        heap, nbytes, offs, zs, zz = _lib.encode_tiled_host(np.ascontiguousarray(a, dtype=a.dtype.newbyteorder("=")), tile, 4,
                                                            blocksize, qcode, dither0, scale)
        # undefined pixels are cfitsio's NULL_VALUE, stated in ZBLANK.  (astropy 4.3.1 checks for nulls neither when it
        # writes nor when it reads quantized images: it turns NaN into INT_MIN on the way in and NULL_VALUE into
        # -2147483647 * ZSCALE + ZZERO on the way out, with or without this card; cfitsio proper and this package read NaN.)
        extra = {"ZQUANTIZ": quantize, "ZDITHER0": int(dither0), "ZBLANK": -2147483647}
    else:
        code = a.dtype.str[1:]
        if code == "u1":
            zbitpix, bytepix, stored = 8, 1, a.astype(np.int32)
        elif code == "i2":
            zbitpix, bytepix, stored = 16, 2, a.astype(np.int32)
        elif code == "u2":
            zbitpix, bytepix, stored = 16, 2, a.astype(np.int32) - 32768
            extra = {"BSCALE": 1, "BZERO": 32768}
        elif code == "i4":
            zbitpix, bytepix, stored = 32, 4, a.astype(np.int32)
        else:
            raise TypeError(f"write_compressed_image: dtype {a.dtype}")
        heap, nbytes, offs, zs, zz = _lib.encode_tiled_host(np.ascontiguousarray(stored), tile, bytepix, blocksize)
    nt = len(nbytes)
    is_float = zs is not None
    row_bytes = 8 + (16 if is_float else 0)
    rows = np.zeros((nt, row_bytes), dtype=np.uint8)
    rows[:, 0:4] = nbytes.astype(">i4").view(np.uint8).reshape(nt, 4)
    rows[:, 4:8] = offs.astype(">i4").view(np.uint8).reshape(nt, 4)
    if is_float:
        rows[:, 8:16] = zs.astype(">f8").view(np.uint8).reshape(nt, 8)
        rows[:, 16:24] = zz.astype(">f8").view(np.uint8).reshape(nt, 8)
    if offs[-1] + nbytes[-1] >= 2 ** 31:
        raise ValueError("heap beyond 2 GiB: 'Q' descriptors are not written")
    cards = [_card("XTENSION", "BINTABLE"), _card("BITPIX", 8), _card("NAXIS", 2), _card("NAXIS1", row_bytes),
             _card("NAXIS2", nt), _card("PCOUNT", int(len(heap))), _card("GCOUNT", 1),
             _card("TFIELDS", 3 if is_float else 1), _card("TTYPE1", "COMPRESSED_DATA"),
             _card("TFORM1", "1PB(%d)" % int(nbytes.max()))]
    if is_float:
        cards += [_card("TTYPE2", "ZSCALE"), _card("TFORM2", "1D"), _card("TTYPE3", "ZZERO"), _card("TFORM3", "1D")]
    cards += [_card("ZIMAGE", True), _card("ZTILE1", tile[0]), _card("ZTILE2", tile[1]), _card("ZCMPTYPE", "RICE_1"),
              _card("ZNAME1", "BLOCKSIZE"), _card("ZVAL1", int(blocksize)), _card("ZNAME2", "BYTEPIX"),
              _card("ZVAL2", bytepix), _card("ZBITPIX", zbitpix), _card("ZNAXIS", 2), _card("ZNAXIS1", nx),
              _card("ZNAXIS2", ny)]
    skip = set(_TABLE_ONLY) | {"SIMPLE", "EXTEND", "BSCALE", "BZERO"} | set(extra)
    for k, v in extra.items():
        cards.append(_card(k, v))
    for k, v in header.items():
        if k in skip or k.startswith("NAXIS") or len(k) > 8:
            continue
        cards.append(_card(k, v))
    cards.append("END".ljust(80))
    blob = "".join(cards).encode("ascii")
    blob += b" " * ((-len(blob)) % BLOCK)
    with open(path, "wb") as f:
        f.write(_header_blob(0, 8, (), primary_header or {}))
        f.write(blob)
        body = rows.tobytes() + heap.tobytes()
        f.write(body + b"\0" * ((-len(body)) % BLOCK))
    return {"scale": scale, "compressed_bytes": int(len(heap)), "n_tiles": int(nt)}


def _copy_range(fi, fo, pos, n):
    """n bytes of file `fi` from offset `pos` appended to `fo` without passing through Python objects (in-kernel copy;
    a reflink where the file system has them)."""
    fo.flush()
    out = fo.tell()
    try:
        done = 0
        while done < n:
            k = os.copy_file_range(fi.fileno(), fo.fileno(), n - done, pos + done, out + done)
            if k == 0:
                raise OSError("short copy")
            done += k
        fo.seek(out + n)
    except (OSError, AttributeError):
        fo.seek(out)
        fi.seek(pos)
        left = n
        while left:
            chunk = fi.read(min(left, 8 << 20))
            if not chunk:
                raise IOError("truncated FITS file")
            fo.write(chunk)
            left -= len(chunk)


def _sum32(buf, start=0):
    """The FITS checksum accumulator (FITS standard 4.0, appendix J): big-endian 32-bit words added in ones' complement
    arithmetic (end-around carry).  `buf`: bytes-like, a multiple of 4 long."""
    a = np.frombuffer(buf, dtype=">u4")
    total = int(start) + int(a.sum(dtype=np.uint64)) if a.size else int(start)
    while total >> 32:
        total = (total & 0xFFFFFFFF) + (total >> 32)
    return total


def _encode_checksum(total):
    """The 16 characters of a CHECKSUM card for an HDU whose words (with the card's value set to sixteen '0') add up to
    `total`: the complement, one byte per four ASCII characters, punctuation avoided, rotated one place to the right
    because the value starts at column 12 of its card (cfitsio `ffesum`, appendix J of the standard)."""
    value = 0xFFFFFFFF - total
    exclude = (0x3A, 0x3B, 0x3C, 0x3D, 0x3E, 0x3F, 0x40, 0x5B, 0x5C, 0x5D, 0x5E, 0x5F, 0x60)
    asc = [0] * 16
    for ii in range(4):
        byte = (value >> (24 - 8 * ii)) & 0xFF
        ch = [byte // 4 + 0x30] * 4
        ch[0] += byte % 4
        check = True
        while check:
            check = False
            for ex in exclude:
                for jj in (0, 2):
                    if ch[jj] == ex or ch[jj + 1] == ex:
                        ch[jj] += 1
                        ch[jj + 1] -= 1
                        check = True
        for jj in range(4):
            asc[4 * jj + ii] = ch[jj]
    return "".join(chr(asc[(ii + 15) % 16]) for ii in range(16))


def _refresh_checksum(blob, datasum=None):
    """A header (bytes, whole blocks) whose CHECKSUM card -- when it has one -- is made valid again for this header and a
    data unit whose words add up to `datasum` (default: what the header's own DATASUM card says; a header without that card
    is returned unchanged unless `datasum` is given).  A DATASUM card is rewritten when `datasum` is given."""
    cards = [blob[i:i + 80] for i in range(0, len(blob), 80)]
    ic = next((i for i, c in enumerate(cards) if c[:9] == b"CHECKSUM="), None)
    idat = next((i for i, c in enumerate(cards) if c[:9] == b"DATASUM ="), None)
    if datasum is not None and idat is not None:
        comment = cards[idat].decode("ascii", "replace")[10:].split("/", 1)
        comment = " /" + comment[1].rstrip() if len(comment) > 1 and "'" not in comment[1] else ""
        cards[idat] = (f"DATASUM = '{int(datasum):<10d}'" + comment)[:80].ljust(80).encode("ascii")
    if ic is None:
        return b"".join(cards)
    if datasum is None:
        if idat is None:
            return blob
        try:
            datasum = int(_parse_value(cards[idat].decode("ascii", "replace")[10:]))
        except (TypeError, ValueError):
            return blob
    tail = cards[ic][29:]  # (the comment: "/ HDU checksum updated ...")
    cards[ic] = b"CHECKSUM= '0000000000000000'" + b" " + tail if len(tail) == 51 else (b"CHECKSUM= '0000000000000000'").ljust(80)
    total = _sum32(b"".join(cards), start=datasum)
    cards[ic] = cards[ic][:11] + _encode_checksum(total).encode("ascii") + cards[ic][27:]
    return b"".join(cards)


def _patch_header(raw, updates, remove=()):
    """The header blocks `raw` (bytes, END card included) with the cards of `updates` (key -> value) replaced in place,
    new keys inserted before END, the cards of `remove` taken out; every other card -- comments, HISTORY, table
    structure, compression keywords -- untouched."""
    cards = [raw[i:i + 80] for i in range(0, len(raw), 80)]
    end = next(i for i, c in enumerate(cards) if c[:8].rstrip() == b"END")
    cards = [c for c in cards[:end] if not (c[8:10] == b"= " and c[:8].decode("ascii", "replace").strip() in remove)]
    left = dict(updates)
    for i, c in enumerate(cards):
        key = c[:8].decode("ascii", "replace").strip()
        if key in left and c[8:10] == b"= ":
            new = _card(key, left.pop(key))
            # keep the card's comment (astropy's header update does): numeric and logical values only -- a '/' inside a
            # string value is not a comment
            old_txt = c.decode("ascii", "replace")
            if len(new) == 80 and "'" not in old_txt[10:] and " /" in old_txt[10:]:
                comment = old_txt[10:].split("/", 1)[1].rstrip()
                body = new.rstrip()
                if len(body) + 3 + len(comment) <= 80:
                    new = (body + " /" + comment).ljust(80)
            cards[i] = new.encode("ascii")
    cards += [_card(k, v).encode("ascii") for k, v in left.items()]
    blob = b"".join(cards) + b"END".ljust(80)
    return blob + b" " * ((-len(blob)) % BLOCK)


def rewrite_with_corrected_headers(path_in, path_out, is_selected, correct):
    """`write_corrected_fits` without touching pixels (utils/Util.py:106-159: every HDU is copied, the selected windows
    get corrected pointing keywords and float32 data).  HDUs that are not selected are copied byte for byte, header
    included.  A selected image HDU whose data unit already holds float32 pixels (BITPIX = -32, no BSCALE / BZERO -- what
    `np.array(data, dtype="<f4")` leaves unchanged) gets a re-written header and its data unit copied as it is; any
    other selected HDU is decoded, converted and encoded as `write_images` does.  `is_selected(i, n_hdus, header)`,
    `correct(header)` (in place).  Returns the number of corrected HDUs."""
    n_corrected = 0
    final = None
    if os.path.exists(path_out) and os.path.samefile(path_in, path_out):  # correcting a file in place
        final, path_out = path_out, str(path_out) + ".coreg-tmp"
    gz_out = None
    if str(final or path_out).endswith(".gz"):  # (astropy's writeto compresses by the name: so does this)
        gz_out, path_out = path_out, str(path_out) + ".coreg-plain"
    path_in = _plain_path(path_in)
    hdus_in, _, raws_in, starts_in = _scan(path_in, with_raw=True)
    with open(path_in, "rb") as fi:
        spans = []
        for hdr, raw_hdr, start in zip(hdus_in, raws_in, starts_in):
            nbytes, shape = _data_size(hdr)
            data_pos = start + len(raw_hdr)
            padded = ((nbytes + BLOCK - 1) // BLOCK) * BLOCK
            spans.append((hdr, start, data_pos, nbytes, padded, shape, raw_hdr))
        size = os.fstat(fi.fileno()).st_size
        with open(path_out, "wb") as fo:
            for i, (hdr, start, data_pos, nbytes, padded, shape, raw_hdr) in enumerate(spans):
                if not is_selected(i, len(spans), hdr):
                    _copy_range(fi, fo, start, min(data_pos + padded, size) - start)
                    continue
                n_corrected += 1
                before = hdr
                hdr = hdr.copy()
                correct(hdr)
                is_image = hdr.get("SIMPLE") is not None or str(hdr.get("XTENSION", "")).strip() == "IMAGE"
                if str(hdr.get("XTENSION", "")).strip() == "BINTABLE" and hdr.get("ZIMAGE") is True:
                    # a tile-compressed image (EUI files): the corrected keywords are patched into the table's header,
                    # the table and its heap -- the compressed pixels -- are copied as they are.  (The reference
                    # decompresses, casts to float32 and compresses again, utils/Util.py:137-138; keeping the original
                    # compressed stream keeps the original pixels.)
                    changed = {k: v for k, v in hdr.items() if k not in before or before[k] != v or
                               type(before[k]) is not type(v)}
                    # (the table and its heap are the input's: DATASUM still holds, CHECKSUM is made valid again)
                    fo.write(_refresh_checksum(_patch_header(raw_hdr, changed)))
                    _copy_range(fi, fo, data_pos, min(padded, size - data_pos))
                    if data_pos + padded > size:
                        fo.write(b"\0" * (data_pos + padded - size))
                    continue
                if not is_image:
                    raise NotImplementedError("only image HDUs (plain or tile-compressed) can be corrected")
                f32 = int(hdr["BITPIX"]) == -32 and hdr.get("BSCALE", 1) == 1 and hdr.get("BZERO", 0) == 0
                changed = {k: v for k, v in hdr.items() if k not in before or before[k] != v or type(before[k]) is not type(v)}
                gone = tuple(k for k in before if k not in hdr)
                if nbytes == 0:
                    fo.write(_refresh_checksum(_patch_header(raw_hdr, changed, gone)))
                elif f32 and int(hdr.get("GCOUNT", 1)) == 1 and int(hdr.get("PCOUNT", 0)) == 0:
                    # the header keeps every card it had (comments, HISTORY, long strings); only the corrected ones change
                    fo.write(_refresh_checksum(_patch_header(raw_hdr, changed, gone)))
                    if data_pos + padded <= size:
                        _copy_range(fi, fo, data_pos, padded)
                    else:  # a file without its final padding
                        _copy_range(fi, fo, data_pos, nbytes)
                        fo.write(b"\0" * (padded - nbytes))
                else:
                    fi.seek(data_pos)
                    data = np.array(_decode(before, fi.read(nbytes), nbytes, shape), dtype="<f4")
                    raw = data.astype(">f4").tobytes()
                    raw += b"\0" * ((-len(raw)) % BLOCK)
                    changed["BITPIX"] = -32
                    blob = _patch_header(raw_hdr, changed, gone + ("BSCALE", "BZERO", "BLANK"))
                    fo.write(_refresh_checksum(blob, datasum=_sum32(raw) if b"DATASUM =" in blob or b"CHECKSUM=" in blob else None))
                    fo.write(raw)
    if gz_out is not None:
        import gzip
        import shutil
        with open(path_out, "rb") as fi, gzip.open(gz_out, "wb", compresslevel=6) as fo:
            shutil.copyfileobj(fi, fo, 1 << 22)
        os.remove(path_out)
        path_out = gz_out
    if final is not None:
        os.replace(path_out, final)
    return n_corrected


def read_all(path):
    """[(data or None, header), ...] for every HDU (used by write_corrected_fits)."""
    return [(d, h) for (h, d) in _read_all(path)]
