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
        flags = mmap.MAP_SHARED | getattr(mmap, "MAP_POPULATE", 0)  # page tables filled now, not by 4096 faults later
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
        hdus, spans = _scan(path)
        i = _select([(h, None) for h in hdus], window)
    except (IOError, KeyError, IndexError, ValueError):
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


def open_cube(path, window=-1):
    """(array, header) of an image HDU of any dimension WITHOUT reading or decoding it: a read-only big-endian view of
    the memory-mapped data unit (NumPy converts what is actually touched, plane by plane, when it is used).  For the
    4-D SPICE windows of which a sweep needs a few wavelength planes.  Falls back to `read_image` (everything decoded)
    for anything that is not a plain, unscaled image HDU of a local file."""
    if isinstance(path, (str, os.PathLike)) and os.path.isfile(path):
        try:
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
                mm = mmap.mmap(f.fileno(), nbytes + (pos - start), flags=mmap.MAP_SHARED | getattr(mmap, "MAP_POPULATE", 0),
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
    return read_image(path, window)


def _scan(path):
    """Headers of every HDU and where each data unit lies: ([header], [(offset, nbytes, shape) or None])."""
    hdus, spans = [], []
    with open(path, "rb") as f:
        while True:
            hdr, _ = _read_header(f)
            if hdr is None:
                break
            nbytes, shape = _data_size(hdr)
            spans.append((f.tell(), nbytes, shape) if nbytes else None)
            f.seek(((nbytes + BLOCK - 1) // BLOCK) * BLOCK, os.SEEK_CUR)
            hdus.append(hdr)
    return hdus, spans


def native_pixels(a):
    """Pixels as the sweep takes them: float32 data (BITPIX=-32, either byte order) as native float32, everything else as
    float64 (the reference's own cast, alignment.py:191 / :198 / :301 / :314; exact for float32)."""
    if isinstance(a, RawImage):
        return a  # uploaded as stored, decoded on the GPU
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
    with open(path_in, "rb") as fi:
        spans = []
        while True:
            start = fi.tell()
            hdr, _ = _read_header(fi)
            if hdr is None:
                break
            nbytes, shape = _data_size(hdr)
            data_pos = fi.tell()
            padded = ((nbytes + BLOCK - 1) // BLOCK) * BLOCK
            fi.seek(padded, os.SEEK_CUR)
            spans.append((hdr, start, data_pos, nbytes, padded, shape))
        size = os.fstat(fi.fileno()).st_size
        with open(path_out, "wb") as fo:
            for i, (hdr, start, data_pos, nbytes, padded, shape) in enumerate(spans):
                if not is_selected(i, len(spans), hdr):
                    _copy_range(fi, fo, start, min(data_pos + padded, size) - start)
                    continue
                n_corrected += 1
                hdr = hdr.copy()
                correct(hdr)
                is_image = hdr.get("SIMPLE") is not None or str(hdr.get("XTENSION", "")).strip() == "IMAGE"
                if "ZIMAGE" in hdr or not is_image:
                    raise NotImplementedError("tile-compressed FITS images need astropy.io.fits (not installed)")
                f32 = int(hdr["BITPIX"]) == -32 and hdr.get("BSCALE", 1) == 1 and hdr.get("BZERO", 0) == 0
                if nbytes == 0:
                    fo.write(_header_blob(i, 8, (), hdr))
                elif f32 and int(hdr.get("GCOUNT", 1)) == 1 and int(hdr.get("PCOUNT", 0)) == 0:
                    fo.write(_header_blob(i, -32, shape, hdr))
                    if data_pos + padded <= size:
                        _copy_range(fi, fo, data_pos, padded)
                    else:  # a file without its final padding
                        _copy_range(fi, fo, data_pos, nbytes)
                        fo.write(b"\0" * (padded - nbytes))
                else:
                    fi.seek(data_pos)
                    data = np.array(_decode(hdr, fi.read(nbytes), nbytes, shape), dtype="<f4")
                    fo.write(_header_blob(i, -32, shape, hdr))
                    raw = data.astype(">f4").tobytes()
                    fo.write(raw + b"\0" * ((-len(raw)) % BLOCK))
    if final is not None:
        os.replace(path_out, final)
    return n_corrected


def read_all(path):
    """[(data or None, header), ...] for every HDU (used by write_corrected_fits)."""
    return [(d, h) for (h, d) in _read_all(path)]
