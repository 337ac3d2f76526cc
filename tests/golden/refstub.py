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
# polars stand-in: tests/golden/refframe.py (numpy columns; filters, group_by, join, concat, ... — see its docstring)
# --------------------------------------------------------------------------
import refframe  # noqa: E402
from refframe import DataFrame, Series, col  # noqa: E402,F401

_Expr = refframe.Expr


def _make_polars():
    pl = types.ModuleType("polars")
    pl.DataFrame = DataFrame
    pl.Series = Series
    pl.col = col
    pl.Expr = _Expr
    for n in ("Utf8", "String", "Int64", "Float64", "Object", "Null", "Boolean"):
        setattr(pl, n, n)
    pl.set_random_seed = lambda seed: None
    pl.lit = refframe.lit
    pl.count = refframe.count
    pl.len = refframe.count
    pl.when = refframe.when
    pl.concat = refframe.concat
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
    # MotifSearchResult subclasses the REAL polars frame and reaches into its internals (motif.py:654-880): replaced by a
    # pass-through with the same column contract (refframe.make_motif_search_result)
    pkg.motif.MotifSearchResult = refframe.make_motif_search_result(pkg.motif.Motif)
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
