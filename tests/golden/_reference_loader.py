"""
Build-container-only loader of the REFERENCE package (`/root/reference/euispice_coreg`) for the golden-vector
generators (`make_golden_alignment.py`, ...).  Nothing of the reference is copied: its modules are imported from where
they lie, under the side interpreter /opt/conda/bin/python3.9 (numpy 1.26.4, scipy 1.7.1, astropy 4.3.1 / wcslib 7.6,
matplotlib), with these LOAD-TIME shims, all stated in DESIGN.md section 1:

  1. `numba`  -> a stub module whose `jit(...)` is the identity decorator.  numba 0.54 fails to initialise against
     numpy 1.26 here; the decorated function (`hdrshift/c_correlate.py:39-72`) is plain sequential float64 Python, so
     running it un-jitted gives the arithmetic numba compiles (same operation order, no fastmath).
  2. `multiprocess.shared_memory` -> the standard library's `multiprocessing.shared_memory` (the package `multiprocess`
     is a fork of it and is not installed for this interpreter; `utils/Util.py:16` only needs `SharedMemory`).
  3. the reference declares python >= 3.11 and writes `list[u.Quantity] | None` in signatures
     (`hdrshift/alignment.py:267-268`), a def-time TypeError on 3.9: a `SourceFileLoader` subclass prepends
     `from __future__ import annotations` to every reference source as it is compiled (annotations become strings;
     no statement of the reference changes).
  4. as `make_golden_rectify.py`: `np.asscalar` / `np.alen` for astropy 4.3.1, and NumPy's NEP-50 ("weak") promotion
     state so that the dtype flow equals the reference's pinned numpy 2.2.6.
  5. `astropy.io.fits.hdu.compressed.compressed` (a sub-module of astropy >= 6, named in an `isinstance` test at
     `utils/Util.py:144`) -> astropy 4.3.1's module `astropy.io.fits.hdu.compressed` itself, which holds the same class.

This file is test infrastructure of the build container; it is never imported by the package, the tests or the bench.
"""
import importlib.abc
import importlib.machinery
import importlib.util
import multiprocessing.shared_memory
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"


def _install_shims():
    for name, fn in [("asscalar", lambda a: a.item()), ("alen", len)]:
        if not hasattr(np, name):
            setattr(np, name, fn)
    np._set_promotion_state("weak")

    numba = types.ModuleType("numba")

    def jit(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        return lambda f: f

    numba.jit = jit
    numba.njit = jit
    sys.modules["numba"] = numba

    mp = types.ModuleType("multiprocess")
    mp.shared_memory = multiprocessing.shared_memory
    sys.modules["multiprocess"] = mp
    sys.modules["multiprocess.shared_memory"] = multiprocessing.shared_memory


class _FutureAnnotationsLoader(importlib.machinery.SourceFileLoader):
    def get_data(self, path):
        data = super().get_data(path)
        if path.endswith(".py"):
            return b"from __future__ import annotations\n" + data
        return data

    def source_to_code(self, data, path, *, _optimize=-1):
        return super().source_to_code(data, path, _optimize=_optimize)

    def get_code(self, fullname):  # never read or write .pyc: the prefix changes line numbers only
        path = self.get_filename(fullname)
        return self.source_to_code(self.get_data(path), path)


class _ReferenceFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path, target=None):
        if fullname != "euispice_coreg" and not fullname.startswith("euispice_coreg."):
            return None
        rel = fullname.split(".")
        base = os.path.join(REFERENCE_ROOT, *rel)
        if os.path.isdir(base):
            init = os.path.join(base, "__init__.py")
            return importlib.util.spec_from_file_location(fullname, init, loader=_FutureAnnotationsLoader(fullname, init),
                                                          submodule_search_locations=[base])
        if os.path.isfile(base + ".py"):
            return importlib.util.spec_from_file_location(fullname, base + ".py",
                                                          loader=_FutureAnnotationsLoader(fullname, base + ".py"))
        return None


def load_reference():
    """Installs the shims and the finder; afterwards `import euispice_coreg.hdrshift.alignment` etc. work."""
    if sys.version_info[:2] != (3, 9) or not os.path.isdir(REFERENCE_ROOT):
        raise SystemExit("run with /opt/conda/bin/python3.9 in the build container (needs /root/reference)")
    sys.dont_write_bytecode = True
    _install_shims()
    if not any(isinstance(f, _ReferenceFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _ReferenceFinder())
    import matplotlib
    matplotlib.use("Agg")
    import astropy.io.fits.hdu.compressed as _comp
    if not hasattr(_comp, "compressed"):
        _comp.compressed = _comp
