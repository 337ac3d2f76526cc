"""Row-list stand-in for the polars calls the reference's POST-PROCESSING and pre-filter code makes (THIS CONTAINER ONLY).

``refstub`` already carries a numpy-backed frame for the scoring path (``filter`` with column predicates, ``get_column``).
The functions behind fixtures g8-g10 — ``postprocess.remove_noisy_motifs / remove_sub_motifs / join_motif_complements``,
``find_motifs_bin.merge_motifs_in_df / process_subpileup / nxgraph_to_dataframe``, ``dataload.filter_pileup /
filter_pileup_minimummod_frequency`` — need a few more frame operations.  This module adds exactly those to the same class,
so that the REAL reference functions run from ``/root/reference`` and only the data-frame container under them is ours.

Polars semantics this stand-in ASSUMES (stated in DESIGN.md §2 as well):
  * nulls: a null percentage is NaN in a float column and ``None`` in an object column; every comparison with a null is
    null and a filter keeps a row only where the predicate is True (Kleene ``|``: null | True = True, null | False = null);
    ``pl.count()`` counts rows whatever they hold, ``(expr).sum()`` skips nulls;
  * ``group_by`` yields its groups in first-appearance order (polars: unspecified; nothing recorded depends on it — every
    recorded table is sorted);
  * ``unique()`` compares Object cells the way py-polars does (its ObjectValue hashes and compares through the Python
    object's ``__hash__`` / ``__eq__``) — identity for ``BetaBernoulliModel``, which defines neither: two rows that hold
    different model OBJECTS are two rows, whatever the counts.  g8-g12 record that no stage of theirs held one motif twice;
    g13 (round 5) records families in which two merge clusters produce the same motif and both rows stay — the product
    follows this since round 5;
  * ``sort`` is stable; a left ``join`` drops the right frame's key columns and suffixes the right frame's other columns
    that clash; ``concat`` is vertical and needs equal column sets.
``nm.motif.MotifSearchResult`` (a subclass of the REAL ``pl.DataFrame`` that reaches into polars internals) is replaced by
``motif_search_result``: required columns checked, derived columns added with the reference's own ``Motif`` methods, the
reference's column order.
"""
from __future__ import annotations

import numpy as np


def _plain(v):
    """str subclasses (the reference's Motif) become plain str, like a Utf8 column would hold them."""
    if isinstance(v, str) and type(v) is not str:
        return str(v)
    return v


def _column(values, n=None):
    """One frame column from a list / array / scalar."""
    if isinstance(values, np.ndarray):
        return values
    if isinstance(values, (str, bytes)) or not hasattr(values, "__len__"):
        values = [values] * (1 if n is None else n)
    vals = [_plain(v) for v in values]
    if vals and all(isinstance(v, (bool, np.bool_)) for v in vals):
        return np.array(vals, dtype=bool)
    if vals and all(isinstance(v, (int, np.integer)) and not isinstance(v, (bool, np.bool_)) for v in vals):
        return np.array(vals, dtype=np.int64)
    if vals and all(isinstance(v, (int, float, np.integer, np.floating)) and not isinstance(v, (bool, np.bool_)) for v in vals):
        return np.array(vals, dtype=np.float64)
    out = np.empty(len(vals), dtype=object)
    for i, v in enumerate(vals):
        out[i] = v
    return out


def _is_null(v):
    return v is None or (isinstance(v, (float, np.floating)) and np.isnan(v))


def _elementwise(a, b, op):
    """Null-aware binary operation on two columns (or a column and a scalar); object result when a null can appear."""
    a_arr = isinstance(a, np.ndarray)
    b_arr = isinstance(b, np.ndarray)
    if (not a_arr or a.dtype != object) and (not b_arr or b.dtype != object) and not (not b_arr and b is None):
        with np.errstate(invalid="ignore"):
            return op(a, b)
    # object columns without a null in them (contig names, strands, mod types of a pileup) take numpy's own loop
    if not (not b_arr and b is None):
        a_null = a_arr and a.dtype == object and bool(np.equal(a, None).any())
        b_null = b_arr and b.dtype == object and bool(np.equal(b, None).any())
        if not a_null and not b_null:
            try:
                res = op(a, b)
                if isinstance(res, np.ndarray) and res.dtype == object and len(res) and isinstance(res[0], (bool, np.bool_)):
                    res = res.astype(bool)
                return res
            except TypeError:
                pass
    n = len(a) if a_arr else len(b)
    out = np.empty(n, dtype=object)
    for i in range(n):
        x = a[i] if a_arr else a
        y = b[i] if b_arr else b
        out[i] = None if (_is_null(x) or _is_null(y)) else op(x, y)
    return out


def _kleene_or(a, b):
    if a.dtype != object and b.dtype != object:
        return a | b
    out = np.empty(len(a), dtype=object)
    for i in range(len(a)):
        x, y = a[i], b[i]
        if x is True or y is True or (x is not None and bool(x)) or (y is not None and bool(y)):
            out[i] = True
        elif x is None or y is None:
            out[i] = None
        else:
            out[i] = False
    return out


def _kleene_and(a, b):
    if a.dtype != object and b.dtype != object:
        return a & b
    out = np.empty(len(a), dtype=object)
    for i in range(len(a)):
        x, y = a[i], b[i]
        fx, fy = (x is not None and not bool(x)), (y is not None and not bool(y))
        if fx or fy:
            out[i] = False
        elif x is None or y is None:
            out[i] = None
        else:
            out[i] = True
    return out


def _truth(mask):
    """Filter semantics: keep where True, drop where False or null."""
    if mask.dtype == object:
        return np.array([m is not None and bool(m) for m in mask], dtype=bool)
    return np.asarray(mask, dtype=bool)


class Expr:
    def __init__(self, fn, name=None):
        self.fn = fn
        self.name = name

    def __call__(self, df):
        return self.fn(df)

    def _bin(self, other, op):
        if isinstance(other, Expr):
            return Expr(lambda df: _elementwise(self(df), other(df), op), self.name)
        return Expr(lambda df: _elementwise(self(df), other, op), self.name)

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

    def __truediv__(self, o):
        return self._bin(o, lambda a, b: a / b)

    def __add__(self, o):
        return self._bin(o, lambda a, b: a + b)

    def __radd__(self, o):
        return Expr(lambda df: _elementwise(o, self(df), lambda a, b: a + b), self.name)

    def __and__(self, o):
        return Expr(lambda df: _kleene_and(np.asarray(self(df)), np.asarray(o(df))), self.name)

    def __or__(self, o):
        return Expr(lambda df: _kleene_or(np.asarray(self(df)), np.asarray(o(df))), self.name)

    def eq(self, o):
        return self == o

    def is_in(self, values):
        vals = [_plain(v) for v in values]

        def fn(df):
            c = self(df)
            pool = set(vals)
            return np.array([x in pool for x in c.tolist()], dtype=bool)
        return Expr(fn, self.name)

    def not_(self):
        def fn(df):
            m = self(df)
            if m.dtype == object:
                return np.array([None if x is None else (not x) for x in m], dtype=object)
            return ~m
        return Expr(fn, self.name)

    def is_null(self):
        return Expr(lambda df: np.array([_is_null(x) for x in self(df).tolist()], dtype=bool), self.name)

    def is_not_null(self):
        return Expr(lambda df: np.array([not _is_null(x) for x in self(df).tolist()], dtype=bool), self.name)

    def alias(self, name):
        return Expr(self.fn, name)

    def sum(self):
        def fn(df):
            v = self(df)
            if v.dtype == object:
                return sum(int(x) for x in v if x is not None)
            return int(np.nansum(v))
        return Expr(fn, self.name)

    def map_elements(self, f, return_dtype=None):
        return Expr(lambda df: _column([None if _is_null(x) else f(x) for x in self(df).tolist()]), self.name)

    def cast(self, dtype):
        return self

    def rolling(self, index_column, *, period, offset=None, closed="right"):
        """``Expr.rolling(index_column=..., period="Ni", offset="-Mi")`` as the polars documentation defines it for an integer
        index column: row i with index value t gets the window ``(t + offset, t + offset + period]`` BY VALUE (closed = "right",
        the default; ``offset`` defaults to ``-period``); the index column must be sorted ascending (polars raises otherwise;
        equal values are allowed and see each other); an expression that is not aggregated yields, per row, the LIST of its
        values over the rows whose index lies in the window, in row order.  Used by the reference's adjacency filter
        (dataload.py:228-247: period "17i", offset "-9i" for adjacency_distance 8 = positions p - 8 .. p + 8)."""
        assert closed == "right", "only the polars default is restated here"

        def span(text):
            assert isinstance(text, str) and text.endswith("i"), f"integer index windows only: {text!r}"
            return int(text[:-1])
        per = span(period)
        off = -per if offset is None else span(offset)
        assert per > 0

        def fn(df):
            idx = np.asarray(df._cols[index_column])
            assert idx.dtype.kind in "iu", "the index column must hold integers"
            idx = idx.astype(np.int64)
            if len(idx) > 1 and (np.diff(idx) < 0).any():
                raise ValueError(f"argument in operation 'rolling' is not sorted: {index_column}")      # polars: InvalidOperationError
            v = self(df)
            lo = np.searchsorted(idx, idx + off, side="right")                # first row with index > t + offset
            hi = np.searchsorted(idx, idx + off + per, side="right")          # one past the last row with index <= t + offset + period
            out = np.empty(len(idx), dtype=object)
            for i in range(len(idx)):
                out[i] = v[lo[i]:hi[i]].tolist()
            return out
        return Expr(fn, self.name)

    @property
    def list(self):
        return _ListNamespace(self)


class _ListNamespace:
    """``Expr.list``: only ``max`` (the largest non-null element of each row's list; null for a list with none)."""

    def __init__(self, expr):
        self.expr = expr

    def max(self):
        def fn(df):
            lists = self.expr(df)
            out = np.full(len(lists), np.nan, dtype=np.float64)
            for i, items in enumerate(lists):
                vals = [x for x in items if not _is_null(x)]
                if vals:
                    out[i] = max(vals)
            return out
        return Expr(fn, self.expr.name)


class _When:
    def __init__(self, cond):
        self.cond = cond

    def then(self, value):
        self.value = value
        return self

    def otherwise(self, other):
        cond, value = self.cond, self.value

        def fn(df):
            c = _truth(np.asarray(cond(df)))
            v = value(df) if isinstance(value, Expr) else _column(value, len(df))
            o = other(df) if isinstance(other, Expr) else _column([other] * len(df))
            out = np.empty(len(df), dtype=object)
            for i in range(len(df)):
                out[i] = v[i] if c[i] else o[i]
            return _column(out.tolist())
        return Expr(fn)


def col(name):
    return Expr(lambda df: df._cols[name], name)


def lit(v):
    return Expr(lambda df: _column(v, len(df)))


def count():
    return Expr(lambda df: len(df), "count")


def when(cond):
    return _When(cond)


class Series:
    def __init__(self, name, values=None, dtype=None):
        self.name = name
        self.values = _column(values if values is not None else [])
        self.dtype = dtype

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


class GroupBy:
    def __init__(self, df, keys):
        self.df, self.keys = df, keys
        order, groups = [], {}
        cols = [df._cols[k].tolist() for k in keys]
        for i, key in enumerate(zip(*cols)):
            if key not in groups:
                groups[key] = []
                order.append(key)
            groups[key].append(i)
        self.groups = [(k, np.array(groups[k], dtype=np.int64)) for k in order]

    def __iter__(self):
        for key, idx in self.groups:
            yield key, self.df._take(idx)

    def map_groups(self, function):
        """Every group's sub-frame (rows in frame order) through ``function``, the results stacked — in first-appearance order of
        the groups here (polars: unspecified without maintain_order; what is recorded from it is sorted)."""
        parts = [function(self.df._take(idx)) for _, idx in self.groups]
        if not parts:
            return self.df._take(np.zeros(0, dtype=np.int64))
        return concat(parts)

    def agg(self, *exprs, **named):
        out = {k: [] for k in self.keys}
        items = [(e.name, e) for e in exprs] + list(named.items())
        for name, _ in items:
            out[name] = []
        for key, idx in self.groups:
            sub = self.df._take(idx)
            for k, v in zip(self.keys, key):
                out[k].append(v)
            for name, e in items:
                out[name].append(e(sub))
        return self.df.__class__(out)


class DataFrame:
    recorder = None        # gen_golden sets a list here: write_motifs appends (path, rows) instead of writing a file

    def __init__(self, data=None, schema=None):
        if isinstance(data, DataFrame):
            self._cols = dict(data._cols)
            return
        data = data or {}
        n = None
        for v in data.values():
            if isinstance(v, np.ndarray) or (hasattr(v, "__len__") and not isinstance(v, (str, bytes))):
                n = len(v)
        self._cols = {k: _column(v, n) for k, v in data.items()}
        lens = {len(v) for v in self._cols.values()}
        assert len(lens) <= 1, f"ragged frame: { {k: len(v) for k, v in self._cols.items()} }"

    # ---- shape
    @property
    def columns(self):
        return list(self._cols)

    def __len__(self):
        if not self._cols:
            return 0
        return len(next(iter(self._cols.values())))

    @property
    def height(self):
        return len(self)

    def is_empty(self):
        return len(self) == 0

    def _like(self, cols):
        out = self.__class__()
        out._cols = cols
        return out

    def _take(self, idx):
        return self._like({k: v[idx] for k, v in self._cols.items()})

    # ---- column access
    def get_column(self, name):
        return Series(name, self._cols[name])

    def __getitem__(self, name):
        return Series(name, self._cols[name])

    # ---- row selection
    def filter(self, *exprs):
        mask = np.ones(len(self), dtype=bool)
        for e in exprs:
            mask &= _truth(np.asarray(e(self)))
        return self._take(mask)

    def remove(self, expr):
        return self._take(~_truth(np.asarray(expr(self))))

    def unique(self):
        seen, keep = set(), []
        cols = [v.tolist() for v in self._cols.values()]
        for i, row in enumerate(zip(*cols)):
            key = tuple(("nan" if isinstance(x, float) and np.isnan(x) else x) for x in row)
            if key not in seen:
                seen.add(key)
                keep.append(i)
        return self._take(np.array(keep, dtype=np.int64))

    def sort(self, by, descending=False):
        by = [by] if isinstance(by, str) else list(by)
        idx = list(range(len(self)))
        for k in reversed(by):
            c = self._cols[k].tolist()
            idx.sort(key=lambda i: c[i], reverse=descending)       # list.sort is stable, also with reverse=True
        return self._take(np.array(idx, dtype=np.int64))

    # ---- columns
    def with_columns(self, *exprs, **named):
        flat = []
        for e in exprs:
            flat += list(e) if isinstance(e, (list, tuple)) else [e]
        cols = dict(self._cols)
        for e in flat:
            if isinstance(e, Series):
                cols[e.name] = e.values
            else:
                assert e.name is not None, "with_columns needs named expressions"
                v = e(self)
                cols[e.name] = v if isinstance(v, np.ndarray) else _column(v, len(self))
        for k, e in named.items():
            cols[k] = e(self)
        return self._like(cols)

    def rename(self, mapping):
        return self._like({mapping.get(k, k): v for k, v in self._cols.items()})

    def select(self, *cols):
        flat = []
        for c in cols:
            flat += list(c) if isinstance(c, (list, tuple)) else [c]
        out = {}
        for c in flat:
            if isinstance(c, Expr):
                out[c.name] = c(self)
            else:
                out[c] = self._cols[c]
        return self._like(out)

    def drop(self, *cols):
        flat = []
        for c in cols:
            flat += list(c) if isinstance(c, (list, tuple)) else [c]
        return self._like({k: v for k, v in self._cols.items() if k not in flat})

    def hstack(self, other):
        cols = dict(self._cols)
        cols.update(other._cols)
        return self._like(cols)

    def group_by(self, *keys):
        flat = []
        for k in keys:
            flat += list(k) if isinstance(k, (list, tuple)) else [k]
        return GroupBy(self, flat)

    def join(self, other, left_on, right_on, how="left", suffix="_right"):
        assert how == "left"
        right_keys = list(zip(*[other._cols[k].tolist() for k in right_on])) if len(other) else []
        index = {}
        for j, key in enumerate(right_keys):
            if not any(_is_null(x) for x in key):
                index.setdefault(key, []).append(j)
        li, ri = [], []
        for i, key in enumerate(zip(*[self._cols[k].tolist() for k in left_on])):
            hits = index.get(key, []) if not any(_is_null(x) for x in key) else []
            if hits:
                for j in hits:
                    li.append(i)
                    ri.append(j)
            else:
                li.append(i)
                ri.append(-1)
        cols = {k: v[np.array(li, dtype=np.int64)] if li else v[:0] for k, v in self._cols.items()}
        for k, v in other._cols.items():
            if k in right_on:
                continue
            name = k + suffix if k in self._cols else k
            vals = [None if j < 0 else v[j] for j in ri]
            cols[name] = _column(vals) if vals else v[:0]
        return self._like(cols)

    # ---- rows out
    def rows(self, columns=None):
        columns = columns or self.columns
        cols = [self._cols[c].tolist() for c in columns]
        return [dict(zip(columns, r)) for r in zip(*cols)]

    def write_motifs(self, path):
        if DataFrame.recorder is not None:
            DataFrame.recorder.append((path, self.rows()))

    def write_csv(self, path, separator=","):
        raise NotImplementedError("the stand-in records tables, it does not write files")


def concat(frames, how="vertical"):
    frames = list(frames)
    assert frames, "concat of nothing"
    names = frames[0].columns
    for f in frames[1:]:
        assert set(f.columns) == set(names), f"concat: column sets differ: {names} vs {f.columns}"
    cols = {}
    for k in names:
        vals = []
        for f in frames:
            vals += f._cols[k].tolist()
        cols[k] = _column(vals)
    out = frames[0].__class__()
    out._cols = cols
    return out


REQUIRED = ["reference", "motif", "mod_type", "mod_position", "model", "score"]
DERIVED = ["n_mod", "n_nomod", "motif_iupac", "mod_position_iupac"]
COMPLEMENTARY = ["motif_complement", "mod_position_complement", "score_complement", "model_complement", "n_mod_complement",
                 "n_nomod_complement", "motif_iupac_complement", "mod_position_iupac_complement"]


def make_motif_search_result(motif_cls):
    """Pass-through for nm.motif.MotifSearchResult (motif.py:654-880): same column contract, none of the polars internals.
    ``motif_cls``: the REFERENCE's Motif (its stripping / IUPAC code derives the columns, motif.py:805-814)."""
    def motif_search_result(data=None, *args, **kwargs):
        df = data if isinstance(data, DataFrame) else DataFrame(data)
        if "contig" in df.columns:
            df = df.rename({"contig": "reference"})
        elif "bin" in df.columns:
            df = df.rename({"bin": "reference"})
        missing = [c for c in REQUIRED if c not in df.columns]
        if missing:
            raise ValueError(f"Missing required columns: {missing}")
        add = {}
        for c in df.columns:
            if c.startswith("model"):
                suffix = c.replace("model", "", 1)
                if f"n_mod{suffix}" not in df.columns or f"n_nomod{suffix}" not in df.columns:
                    ms = df._cols[c].tolist()
                    add[f"n_mod{suffix}"] = [None if m is None else m._alpha - m._alpha_prior for m in ms]
                    add[f"n_nomod{suffix}"] = [None if m is None else m._beta - m._beta_prior for m in ms]
        if "motif_iupac" not in df.columns and len(df):
            ms = [motif_cls(s, p).new_stripped_motif() for s, p in zip(df._cols["motif"].tolist(), df._cols["mod_position"].tolist())]
            add["mod_position_iupac"] = [m.mod_position for m in ms]
            add["motif_iupac"] = [m.iupac() for m in ms]
        if add:
            df = df.hstack(DataFrame(add))
        fixed = REQUIRED + DERIVED + (COMPLEMENTARY if set(COMPLEMENTARY) & set(df.columns) else [])
        order = [c for c in fixed if c in df.columns] + [c for c in df.columns if c not in fixed]
        return df.select(order)
    return motif_search_result
