"""Import harness for the upstream reference (THIS CONTAINER ONLY).

Used only by ``tests/golden/gen_golden.py`` to run the real reference code and
record input/output vectors as fixtures.  ``/root/reference`` does not exist on
the GPU box, so nothing in ``tests/``'s test functions, ``smoke()`` or
``bench.py`` imports this module.

The reference package cannot be imported as-is here: ``nanomotif/__init__.py``
eagerly imports polars / pysam / pyfastx / epymetheus, none of which are
installed.  Instead we register an empty ``nanomotif`` package whose
``__path__`` points at the reference checkout (skipping its ``__init__``),
inject empty stubs for the I/O-only dependencies, and provide a tiny
numpy-backed stand-in for the handful of polars operations the scoring path
uses (``filter`` with column predicates, ``get_column``, ``unique``,
``to_numpy``/``to_list``).  No reference source is copied; the modules are
imported from where they lie.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("NANOMOTIF_REFERENCE", "/root/reference")


# --------------------------------------------------------------------------
# minimal polars stand-in
# --------------------------------------------------------------------------
class _Expr:
    def __init__(self, fn):
        self.fn = fn

    def __call__(self, df):
        return self.fn(df)

    def _bin(self, other, op):
        if isinstance(other, _Expr):
            return _Expr(lambda df: op(self(df), other(df)))
        return _Expr(lambda df: op(self(df), other))

    def __ge__(self, o):
        return self._bin(o, lambda a, b: a >= b)

    def __le__(self, o):
        return self._bin(o, lambda a, b: a <= b)

    def __gt__(self, o):
        return self._bin(o, lambda a, b: a > b)

    def __lt__(self, o):
        return self._bin(o, lambda a, b: a < b)

    def __eq__(self, o):  # type: ignore[override]
        return self._bin(o, lambda a, b: a == b)

    def __and__(self, o):
        return self._bin(o, lambda a, b: a & b)

    def __or__(self, o):
        return self._bin(o, lambda a, b: a | b)

    def eq(self, o):
        return self == o

    def is_in(self, values):
        vals = list(values)
        return _Expr(lambda df: np.isin(self(df), np.array(vals, dtype=object)
                                         if vals and isinstance(vals[0], str) else np.array(vals)))

    def not_(self):
        return _Expr(lambda df: ~self(df))


def col(name):
    return _Expr(lambda df: df._cols[name])


class Series:
    def __init__(self, name, values):
        self.name = name
        self.values = np.asarray(values)

    def to_numpy(self):
        return self.values

    def to_list(self):
        return self.values.tolist()

    def unique(self):
        # polars' unique() order is unspecified; the fixtures pin "sorted"
        return Series(self.name, np.unique(self.values))

    def __iter__(self):
        return iter(self.values.tolist())

    def __len__(self):
        return len(self.values)

    def __getitem__(self, i):
        return self.values[i]


class DataFrame:
    def __init__(self, data=None, schema=None):
        data = data or {}
        self._cols = {k: (np.asarray(v, dtype=object) if len(v) and isinstance(v[0], str) else np.asarray(v))
                      for k, v in data.items()}

    @property
    def columns(self):
        return list(self._cols)

    def filter(self, expr):
        mask = np.asarray(expr(self), dtype=bool)
        out = DataFrame()
        out._cols = {k: v[mask] for k, v in self._cols.items()}
        return out

    def get_column(self, name):
        return Series(name, self._cols[name])

    def __getitem__(self, name):
        return Series(name, self._cols[name])

    def is_empty(self):
        return len(self) == 0

    def __len__(self):
        if not self._cols:
            return 0
        return len(next(iter(self._cols.values())))

    @property
    def height(self):
        return len(self)


def _make_polars():
    pl = types.ModuleType("polars")
    pl.DataFrame = DataFrame
    pl.Series = Series
    pl.col = col
    pl.Expr = _Expr
    for n in ("Utf8", "String", "Int64", "Float64", "Object", "Null", "Boolean"):
        setattr(pl, n, n)
    pl.set_random_seed = lambda seed: None
    pl.lit = lambda v: _Expr(lambda df: np.full(len(df), v))
    testing = types.ModuleType("polars.testing")
    pl.testing = testing
    return pl, testing


_loaded = None


def load_reference():
    """Return the stub ``nanomotif`` package with the hot-path modules imported."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not os.path.isdir(os.path.join(REFERENCE_ROOT, "nanomotif")):
        raise RuntimeError(f"reference checkout not found at {REFERENCE_ROOT}")

    pl, pl_testing = _make_polars()
    sys.modules["polars"] = pl
    sys.modules["polars.testing"] = pl_testing
    for name in ("pysam", "progressbar", "pyfastx", "epymetheus"):
        sys.modules.setdefault(name, types.ModuleType(name))
    pyinstr = types.ModuleType("pyinstrument")
    pyinstr.Profiler = object
    sys.modules.setdefault("pyinstrument", pyinstr)
    sys.modules["epymetheus"].query_pileup_records = None
    sys.modules["epymetheus"].PileupColumn = None

    pkg = types.ModuleType("nanomotif")
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "nanomotif")]
    sys.modules["nanomotif"] = pkg
    for mod in ("constants", "model", "utils", "seq", "motif", "logger", "parallel", "seed",
                "postprocess", "find_motifs_bin"):
        m = importlib.import_module(f"nanomotif.{mod}")
        setattr(pkg, mod, m)
    _loaded = pkg
    return pkg


def make_pileup(contig, position, strand, fraction_mod, mod_type=None, nvalid=None):
    """Build a stand-in polars frame with the reference's pileup column names."""
    n = len(position)
    data = {
        "contig": np.asarray(contig, dtype=object),
        "position": np.asarray(position, dtype=np.int64),
        "strand": np.asarray(strand, dtype=object),
        "fraction_mod": np.asarray(fraction_mod, dtype=np.float64),
    }
    if mod_type is not None:
        data["mod_type"] = np.asarray(mod_type, dtype=object)
    if nvalid is not None:
        data["Nvalid_cov"] = np.asarray(nvalid, dtype=np.int64)
    df = DataFrame()
    df._cols = data
    assert all(len(v) == n for v in data.values())
    return df
