"""
Minimal FITS image I/O for the drop-in `Alignment` API (the reference uses astropy.io.fits, which is not a
dependency here).  Reads / writes primary and IMAGE-extension HDUs (BITPIX 8/16/32/64/-32/-64, BSCALE/BZERO);
headers are plain insertion-ordered dicts.  Tile-compressed images (ZIMAGE binary tables) need astropy: when
astropy is importable it is used transparently, otherwise a clear error is raised.
"""
from __future__ import annotations

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
    if s.startswith("'"):
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
    s = s.split("/")[0].strip()
    if s in ("T", "F"):
        return s == "T"
    try:
        return int(s)
    except ValueError:
        pass
    try:
        return float(s.replace("D", "E").replace("d", "e"))
    except ValueError:
        return s


def _read_header(f):
    hdr = Header()
    raw = b""
    while True:
        block = f.read(BLOCK)
        if len(block) < BLOCK:
            if not raw and not block:
                return None, b""
            raise IOError("truncated FITS header")
        raw += block
        done = False
        for i in range(0, BLOCK, 80):
            card = block[i:i + 80].decode("ascii", "replace")
            key = card[:8].strip()
            if key == "END":
                done = True
                break
            if not key or key in ("COMMENT", "HISTORY") or card[8:10] != "= ":
                continue
            hdr[key] = _parse_value(card[10:])
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


def _read_all(path, with_data=True, only=None):
    """[(header, data)] of every HDU.  with_data=False: headers only (data skipped with a seek); only=<window>: the pixel
    data of that HDU alone are read and decoded (an index, or an EXTNAME)."""
    hdus = []
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


def native_pixels(a):
    """Pixels as the sweep takes them: float32 data (BITPIX=-32, either byte order) as native float32, everything else as
    float64 (the reference's own cast, alignment.py:191 / :198 / :301 / :314; exact for float32)."""
    a = np.asarray(a)
    if a.dtype.kind == "f" and a.dtype.itemsize == 4:
        return np.ascontiguousarray(a, dtype=np.float32)
    return np.array(a, dtype=np.float64)


def read_image(path, window=-1):
    """(data, header) of one HDU.  `path` may also be a (data, header) pair already in memory."""
    if isinstance(path, (tuple, list)) and len(path) == 2:
        return np.asarray(path[0]), Header(path[1])
    try:
        import astropy.io.fits as afits  # optional: compressed images, remote URLs
    except ImportError:
        afits = None
    if afits is not None:
        with afits.open(path) as hl:
            hdu = hl[window]
            return np.array(hdu.data), Header({k: hdu.header[k] for k in hdu.header.keys() if k})
    if not os.path.exists(str(path)):
        raise FileNotFoundError(path)
    hdus = _read_all(path, only=window)
    hdr, data = hdus[_select(hdus, window)]
    if data is None:
        if "ZIMAGE" in hdr or str(hdr.get("XTENSION", "")).strip() == "BINTABLE":
            raise NotImplementedError("tile-compressed FITS images need astropy.io.fits (not installed)")
        raise ValueError(f"HDU {window!r} of {path} holds no image")
    return data, hdr


def read_header(path, window=-1):
    """Header of one HDU without decoding any pixel data."""
    if isinstance(path, (tuple, list)) and len(path) == 2:
        return Header(path[1])
    try:
        import astropy.io.fits as afits
    except ImportError:
        afits = None
    if afits is not None:
        with afits.open(path) as hl:
            hdu = hl[window]
            return Header({k: hdu.header[k] for k in hdu.header.keys() if k})
    if not os.path.exists(str(path)):
        raise FileNotFoundError(path)
    hdus = _read_all(path, with_data=False)
    return hdus[_select(hdus, window)][0]


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
        body = "'" + f"{sv:<8}" + "'"
    return f"{key:<8}= {body}"[:80].ljust(80)


def write_images(path, hdus, overwrite=True):
    """hdus: list of (data or None, header dict).  The first is written as the primary HDU."""
    if os.path.exists(path) and not overwrite:
        raise OSError(f"{path} exists")
    structural = ("SIMPLE", "XTENSION", "BITPIX", "NAXIS", "EXTEND", "PCOUNT", "GCOUNT", "BSCALE", "BZERO")
    with open(path, "wb") as f:
        for i, (data, hdr) in enumerate(hdus):
            cards = []
            if data is None:
                bitpix, shape, raw = 8, (), b""
            else:
                data = np.asarray(data)
                bitpix = {"u1": 8, "i2": 16, "i4": 32, "i8": 64, "f4": -32, "f8": -64}[data.dtype.str[1:]]
                shape = data.shape
                raw = data.astype(_BITPIX_DTYPE[bitpix]).tobytes()
            cards.append(_card("SIMPLE", True) if i == 0 else _card("XTENSION", "IMAGE"))
            cards.append(_card("BITPIX", bitpix))
            cards.append(_card("NAXIS", len(shape)))
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
            blob += b" " * ((-len(blob)) % BLOCK)
            f.write(blob)
            if raw:
                f.write(raw + b"\0" * ((-len(raw)) % BLOCK))


def read_all(path):
    """[(data or None, header), ...] for every HDU (used by write_corrected_fits)."""
    return [(d, h) for (h, d) in _read_all(path)]
