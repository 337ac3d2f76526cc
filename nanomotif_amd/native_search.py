"""Python face of the native lock-step search (csrc/nmsearch.cpp, include/nmscan.h: nm_search_*).

``find_best_candidates_all`` is what ``run_lockstep`` over ``search.find_best_candidates_co`` coroutines computes —
(graph, best candidates, background PSSM) per (bin, mod type) task, reference: find_motifs_bin.py:606-839 — with the
whole state machine of every task running inside libnmscan: per round one window batch and one scoring batch on the
engine, no interpreter in the loop.  ``search.py`` stays as the readable twin (CPU tests drive both against the same
recorded traces)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from .model import BetaBernoulliModel
from .motif import MOD_TYPE_TO_CANONICAL, Motif
from .search import MotifTree


def _params(padding, min_kl, score_threshold, max_dead_ends=25, max_rounds_since_new_best=30, remaining_sequences_threshold=0.001,
            freq_threshold=0.15, max_motif_length=25):
    return _lib.SearchParams(int(padding), int(max_dead_ends), int(max_rounds_since_new_best), int(max_motif_length),
                             float(min_kl), float(score_threshold), float(remaining_sequences_threshold), float(freq_threshold))


class SearchResults:
    """Per-task results of one native run; graphs are materialised on demand (only the CLI's GML export and the tests
    look at more than the best candidates)."""

    def __init__(self, lib, handle, keys, mod_types, pssms, padding):
        self.keys, self.mod_types, self.pssms, self.padding = list(keys), list(mod_types), pssms, int(padding)
        W = 2 * self.padding + 1
        n = len(self.keys)
        nn, ne, nb = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        stats = (C.c_uint64 * 3)()
        _lib.check(lib.nm_search_result_sizes(handle, C.byref(nn), C.byref(ne), C.byref(nb), stats))
        self.rounds, self.candidates, self.window_requests = (int(x) for x in stats)
        _lib.check(lib.nm_search_result_speculation(handle, stats))
        self.iterations, self.spec_hits, self.spec_misses = (int(x) for x in stats)
        self.node_off, self.edge_off, self.best_off = (np.zeros(n + 1, dtype=np.uint64) for _ in range(3))
        self.none = np.zeros(max(n, 1), dtype=np.uint8)
        self.motif = np.zeros(max(int(nn.value) * W, 1), dtype=np.uint8)
        self.counts = np.zeros((max(int(nn.value), 1), 2), dtype=np.int64)
        self.score = np.zeros(max(int(nn.value), 1), dtype=np.float64)
        self.priority = np.zeros(max(int(nn.value), 1), dtype=np.float64)
        self.depth = np.zeros(max(int(nn.value), 1), dtype=np.int32)
        self.visited = np.zeros(max(int(nn.value), 1), dtype=np.uint8)
        self.edges = np.zeros((max(int(ne.value), 1), 2), dtype=np.int32)
        self.best = np.zeros(max(int(nb.value), 1), dtype=np.int32)
        p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
        _lib.check(lib.nm_search_result_export(handle, p(self.node_off, C.c_uint64), p(self.edge_off, C.c_uint64), p(self.best_off, C.c_uint64),
                                               p(self.none, C.c_uint8), self.motif.ctypes.data_as(C.c_char_p), p(self.counts, C.c_int64),
                                               p(self.score, C.c_double), p(self.priority, C.c_double), p(self.depth, C.c_int32),
                                               p(self.visited, C.c_uint8), p(self.edges, C.c_int32), p(self.best, C.c_int32)))
        self._lib, self._handle = lib, handle              # kept for post-processing (nm_post_run reads the graphs); close() frees it
        import threading
        self._close_lock = threading.Lock()
        self.W = W
        self._text = self.motif.tobytes().decode("ascii") if nn.value else ""

    def close(self):
        with self._close_lock:                      # (discover() frees the graphs on a side thread; __del__ may come from another)
            handle, self._handle = self._handle, None
        if handle is not None:
            self._lib.nm_search_result_free(handle)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def postprocess(self, engine, task_bins, merge_slots, reduce=None, tables=False):
        """process_subpileup after the search for every task (nm_post_run): ``task_bins`` / ``merge_slots`` = engine bin
        index and the classification the merge stage is scored on, per task.  Returns PostResults."""
        n = len(self.keys)
        bins = np.ascontiguousarray(np.fromiter(task_bins, dtype=np.uint32, count=n)) if n else np.zeros(1, np.uint32)
        slots = np.ascontiguousarray(np.fromiter(merge_slots, dtype=np.uint32, count=n)) if n else np.zeros(1, np.uint32)
        err = []
        cb = _reduce_callback(reduce, err)
        handle = C.c_void_p()
        rc = self._lib.nm_post_run(engine.ctx, self._handle, bins.ctypes.data_as(C.POINTER(C.c_uint32)), slots.ctypes.data_as(C.POINTER(C.c_uint32)),
                                   cb, None, C.byref(handle))
        if err:
            raise err[0]
        _lib.check(rc)
        return PostResults(self._lib, handle, self.keys, tables=tables)

    def postprocess_custom(self, score_fn, tables=False):
        """The same on a Python scorer (CPU tests): ``score_fn(list of (task index, Motif)) -> int64[n, 2]``, every request
        on the merge stage's 0.3 / 0.7 classification."""
        err = []
        handle = C.c_void_p()
        rc = self._lib.nm_post_run_custom(self._handle, _post_score_callback(score_fn, err), None, C.byref(handle))
        if err:
            raise err[0]
        _lib.check(rc)
        return PostResults(self._lib, handle, self.keys, tables=tables)

    def _node(self, t, k):
        i = int(self.node_off[t]) + k
        m = Motif(self._text[i * self.W:(i + 1) * self.W], self.padding)
        attrs = dict(model=BetaBernoulliModel.from_counts(*self.counts[i]), motif=m, visited=bool(self.visited[i]), score=float(self.score[i]),
                     priority=(0 if self.priority[i] == 0 and self.depth[i] == 0 else float(self.priority[i])), depth=int(self.depth[i]))
        return m, attrs

    def artifacts(self, t):
        """What ``find_motifs_bin.write_search_artifacts`` needs of task ``t`` — (something with ``gml_text()``, None, background PSSM),
        or None when the search found nothing — without building the graph's Python objects: the text is
        ``result(t, full_graph=True)[0].gml_text()`` byte for byte (nodes in node order; a node's edges in the order they were made)."""
        if self.none[t]:
            return None
        if os.environ.get("NANOMOTIF_PY_GML") != "1":
            # the text comes from the library, all tasks in one call (nm_search_result_gml; NANOMOTIF_PY_GML=1: built here, as before round 6)
            if getattr(self, "_gml_native", None) is None:
                text, off, n_off = C.c_void_p(), C.POINTER(C.c_uint64)(), C.c_uint64(0)
                with self._close_lock:
                    if self._handle is None:
                        raise RuntimeError("the search result has been freed")
                    _lib.check(self._lib.nm_search_result_gml(self._handle, C.byref(text), C.byref(off), C.byref(n_off)))
                    offs = np.ctypeslib.as_array(off, shape=(int(n_off.value),)).tolist()
                    self._gml_native = (C.string_at(text, offs[-1]), offs)
            blob, offs = self._gml_native
            return _GmlText(blob[offs[t]:offs[t + 1]].decode("ascii")), None, self.pssms[t]
        if getattr(self, "_gml_cols", None) is None:
            self._gml_cols = (self.score.tolist(), self.priority.tolist(), self.depth.tolist(), self.visited.tolist())
        score, priority, depth, visited = self._gml_cols
        a, b = int(self.node_off[t]), int(self.node_off[t + 1])
        W, text = self.W, self._text
        out = ["graph [\n  directed 1\n"]
        for k in range(b - a):
            i = a + k
            label = text[i * W:(i + 1) * W].strip(".")
            out.append(f'  node [\n    id {k}\n    label "{label}"\n    score {score[i]}\n    priority {priority[i]}\n    depth {depth[i]}\n'
                       f'    visited {1 if visited[i] else 0}\n  ]\n')
        e = self.edges[int(self.edge_off[t]):int(self.edge_off[t + 1])]
        if len(e):
            e = e[np.argsort(e[:, 0], kind="stable")]
            for u, v in e.tolist():
                out.append(f"  edge [\n    source {u}\n    target {v}\n  ]\n")
        out.append("]\n")
        return _GmlText("".join(out)), None, self.pssms[t]

    def result(self, t, full_graph=False):
        """(graph, best, bin_pssm) of task ``t`` or None, as ``find_best_candidates_co`` returns it.  Without
        ``full_graph`` the graph holds the best candidates only (all that post-processing reads)."""
        if self.none[t]:
            return None
        g = MotifTree()
        a, b = int(self.node_off[t]), int(self.node_off[t + 1])
        best_idx = self.best[int(self.best_off[t]):int(self.best_off[t + 1])].tolist()
        wanted = range(b - a) if full_graph else sorted(set(best_idx))
        made = {}
        for k in wanted:
            m, attrs = self._node(t, k)
            g.add_node(m, **attrs)
            made[k] = m
        if full_graph:
            for u, v in self.edges[int(self.edge_off[t]):int(self.edge_off[t + 1])].tolist():
                g.add_edge(made[u], made[v])
        return g, [made[k] for k in best_idx], self.pssms[t]


class _GmlText:
    def __init__(self, text):
        self._text = text

    def gml_text(self):
        return self._text


def _reduce_callback(reduce, err):
    if reduce is None:
        return C.cast(None, _lib.SEARCH_REDUCE_FN)

    def _reduce(_user, ptr, count):
        try:
            a = np.ctypeslib.as_array(ptr, shape=(int(count),))
            a[:] = reduce(a.copy())
            return 0
        except Exception as e:          # a Python exception cannot cross the C frames
            err.append(e)
            return -3
    return _lib.SEARCH_REDUCE_FN(_reduce)


def _post_score_callback(score_fn, err):
    def _score(_user, cnt, task, text, off, modpos, out):
        try:
            s = C.string_at(text, off[cnt]).decode("ascii") if cnt else ""
            res = np.asarray(score_fn([(int(task[i]), Motif(s[off[i]:off[i + 1]], int(modpos[i]))) for i in range(cnt)]), dtype=np.int64)
            np.ctypeslib.as_array(out, shape=(cnt, 2))[:] = res
            return 0
        except Exception as e:
            err.append(e)
            return -3
    return _lib.POST_SCORE_FN(_score)


def postprocess_rows_custom(keys, rows_per_task, padding, score_fn, tables=False):
    """Native post-processing of explicit rows (tests): ``rows_per_task[t]`` = list of (motif string of 2 * padding + 1
    characters, n_mod, n_nomod, score) in graph node order.  Returns PostResults."""
    lib = _lib.load()
    W = 2 * int(padding) + 1
    off = np.zeros(len(keys) + 1, dtype=np.uint64)
    np.cumsum([len(r) for r in rows_per_task], out=off[1:])
    flat = [r for rows in rows_per_task for r in rows]
    text = "".join(r[0] for r in flat).encode("ascii")
    assert len(text) == len(flat) * W
    counts = np.ascontiguousarray(np.array([[r[1], r[2]] for r in flat], dtype=np.int64).reshape(-1, 2))
    score = np.ascontiguousarray(np.array([r[3] for r in flat], dtype=np.float64))
    err = []
    handle = C.c_void_p()
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    rc = lib.nm_post_run_rows_custom(len(keys), W, p(off, C.c_uint64), text, p(counts, C.c_int64) if len(flat) else None,
                                     p(score, C.c_double) if len(flat) else None, _post_score_callback(score_fn, err), None, C.byref(handle))
    if err:
        raise err[0]
    _lib.check(rc)
    return PostResults(lib, handle, keys, tables=tables)


class PostResults:
    """Rows of every (task, stage) of one native post-processing run (include/nmscan.h: nm_post_*), as the MotifRow
    records postprocess.postprocess_co produces; built on demand (a run without --out only reads the last stage)."""
    STAGES = ("motifs", "motifs-noise", "motifs-noise-merge", "motifs-noise-merge-sub", "motifs-noise-merge-sub-complement")

    def __init__(self, lib, handle, keys, tables=False):
        """``tables``: also keep every (task, stage) table as the TEXT ``postprocess.format_motifs`` would write (nm_post_tables) — a run
        with --out writes five of them per task and needs no row objects for the first four."""
        self.keys = list(keys)
        self._tables = None
        if tables and self.keys:
            n_t = len(self.keys)
            ref = (C.c_char_p * n_t)(*[str(k[0]).encode() for k in self.keys])
            mod = (C.c_char_p * n_t)(*[str(k[1]).encode() for k in self.keys])
            text, off, n_off = C.c_void_p(), C.POINTER(C.c_uint64)(), C.c_uint64(0)
            _lib.check(lib.nm_post_tables(handle, ref, mod, C.byref(text), C.byref(off), C.byref(n_off)))
            offs = np.ctypeslib.as_array(off, shape=(int(n_off.value),)).tolist()
            self._tables = (C.string_at(text, offs[-1]), offs)                        # (bytes: the offsets count bytes, a bin name may not be ASCII)
        nr, nb = C.c_uint64(0), C.c_uint64(0)
        stats = (C.c_uint64 * 2)()
        _lib.check(lib.nm_post_sizes(handle, C.byref(nr), C.byref(nb), stats))
        self.batches, self.candidates = int(stats[0]), int(stats[1])
        n = int(nr.value)
        self.task = np.zeros(max(n, 1), dtype=np.uint32)
        self.stage = np.zeros(max(n, 1), dtype=np.uint8)
        self.text_off = np.zeros(2 * n + 1, dtype=np.uint64)
        text = np.zeros(max(int(nb.value), 1), dtype=np.uint8)
        self.mod_position = np.zeros(max(n, 1), dtype=np.int32)
        self.mod_position_iupac = np.zeros(max(n, 1), dtype=np.int32)
        self.counts = np.zeros((max(n, 1), 2), dtype=np.int64)
        self.score = np.zeros(max(n, 1), dtype=np.float64)
        self.complement = np.zeros(max(n, 1), dtype=np.int64)
        p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
        _lib.check(lib.nm_post_export(handle, p(self.task, C.c_uint32), p(self.stage, C.c_uint8), p(self.text_off, C.c_uint64),
                                      text.ctypes.data_as(C.c_char_p), p(self.mod_position, C.c_int32), p(self.mod_position_iupac, C.c_int32),
                                      p(self.counts, C.c_int64), p(self.score, C.c_double), p(self.complement, C.c_int64)))
        lib.nm_post_free(handle)
        self.n = n
        self._text = text.tobytes().decode("ascii")
        key = self.task[:n].astype(np.int64) * 8 + self.stage[:n]
        self._first = np.searchsorted(key, np.arange(len(self.keys) * 8 + 1)).tolist()       # records are sorted by (task, stage)
        self._off = self.text_off.tolist()
        # plain lists: a row is built from a dozen scalars, and numpy scalars cost more to unwrap than the row costs to make
        self._cols = (self.task[:n].tolist(), self.stage[:n].tolist(), self.mod_position[:n].tolist(), self.mod_position_iupac[:n].tolist(),
                      self.counts[:n].tolist(), self.score[:n].tolist(), self.complement[:n].tolist())
        self._made = {}
        from .postprocess import MotifRow
        self._MotifRow = MotifRow

    def _row(self, i):
        r = self._made.get(i)
        if r is None:
            MotifRow = self._MotifRow                       # (bound once: an import statement per row costs a third of making the row)
            task, stage, modpos, modpos_iu, counts, score, complement = self._cols
            k = self.keys[task[i]]
            o, text = self._off, self._text
            comp = complement[i]
            # (the fields set directly: a dataclass __init__ and a classmethod per row are half of what a row costs, and a 1 Gbp run
            #  makes 2 600 of them; the counts are Python ints already — .tolist() —, like update() gets them)
            model = BetaBernoulliModel.__new__(BetaBernoulliModel)
            model._alpha_prior = model._beta_prior = 5
            model._alpha, model._beta = 5 + counts[i][0], 5 + counts[i][1]
            r = MotifRow.__new__(MotifRow)
            r.__dict__ = {"reference": k[0], "motif": text[o[2 * i]:o[2 * i + 1]], "mod_type": k[1], "mod_position": modpos[i], "model": model,
                          "score": score[i], "complement": None if comp < 0 else self._row(comp), "has_complement_columns": stage[i] == 4,
                          "_cache": (text[o[2 * i + 1]:o[2 * i + 2]], modpos_iu[i], None)}       # (the reverse complement on first use)
            self._made[i] = r
        return r

    def table_text(self, t, stage):
        """The text of table (task ``t``, ``stage``) — ``format_motifs(self.rows(t, stage))`` — or None when the tables were not asked for."""
        if self._tables is None:
            return None
        text, off = self._tables
        return text[off[t * 5 + stage]:off[t * 5 + stage + 1]].decode("utf-8")

    def n_stages(self, t):
        """Stages of task ``t`` that hold rows (the reference stops a task at the first empty one)."""
        f = self._first
        return sum(1 for s in range(5) if f[t * 8 + s + 1] > f[t * 8 + s])

    def rows(self, t, stage):
        f = self._first
        return [self._row(i) for i in range(f[t * 8 + stage], f[t * 8 + stage + 1])]

    def final(self, t):
        """What postprocess_co returns for task ``t``: the rows of the last stage, or None."""
        return self.rows(t, 4) or None


def find_best_candidates_all(engine, tasks, padding, min_kl, score_threshold, reduce=None, **kw):
    """tasks: list of (key=(bin name, mod type), window-store task id, total windows, background PSSM float64[4, W]).
    Runs every search on ``engine`` (nm_search_run); ``reduce``: callable summing an int64 numpy array over the ranks of a
    contig-sharded run (None on one GPU).  Returns SearchResults."""
    lib = engine.lib
    n = len(tasks)
    W = 2 * int(padding) + 1
    keys = [t[0] for t in tasks]
    mods = [k[1] for k in keys]
    u32 = lambda xs: np.ascontiguousarray(np.fromiter(xs, dtype=np.uint32, count=n))
    bins = u32(engine.bin_index[k[0]] for k in keys)
    slots = u32(engine.slot_of_mod[k[1]] for k in keys)
    wins = u32(t[1] for t in tasks)
    totals = np.ascontiguousarray(np.fromiter((t[2] for t in tasks), dtype=np.uint64, count=n))
    canon = np.frombuffer("".join(MOD_TYPE_TO_CANONICAL[m] for m in mods).encode("ascii"), dtype=np.uint8).copy() if n else np.zeros(1, np.uint8)
    pssm = np.ascontiguousarray(np.stack([np.asarray(t[3], dtype=np.float64).reshape(4, W) for t in tasks])) if n else np.zeros((1, 4, W))
    err = []
    cb = _reduce_callback(reduce, err)
    params = _params(padding, min_kl, score_threshold, **kw)
    handle = C.c_void_p()
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    rc = lib.nm_search_run(engine.ctx, n, p(bins, C.c_uint32), p(slots, C.c_uint32), p(wins, C.c_uint32), C.byref(params),
                           p(pssm, C.c_double), p(totals, C.c_uint64), p(canon, C.c_uint8), cb, None, C.byref(handle))
    if err:
        raise err[0]
    _lib.check(rc)
    return SearchResults(lib, handle, keys, mods, [t[3] for t in tasks], padding)


def find_best_candidates_custom(tasks, padding, min_kl, score_threshold, score_fn, window_fn, **kw):
    """The same state machine on Python back ends (CPU tests): ``score_fn(list of (task index, Motif)) -> int64[n, 2]``,
    ``window_fn(list of (task index, kind, Motif)) -> int32[n, 258]`` (kind 'pssm' / 'remove', rows like nm_win_batch).
    tasks: list of (key, total windows, background PSSM)."""
    lib = _lib.load()
    n = len(tasks)
    W = 2 * int(padding) + 1
    keys = [t[0] for t in tasks]
    mods = [k[1] for k in keys]
    totals = np.ascontiguousarray(np.fromiter((t[1] for t in tasks), dtype=np.uint64, count=n))
    canon = np.frombuffer("".join(MOD_TYPE_TO_CANONICAL[m] for m in mods).encode("ascii"), dtype=np.uint8).copy()
    pssm = np.ascontiguousarray(np.stack([np.asarray(t[2], dtype=np.float64).reshape(4, W) for t in tasks]))
    err = []

    def _score(_user, cnt, task, motifs, out):
        try:
            text = C.string_at(motifs, cnt * W).decode("ascii")
            res = np.asarray(score_fn([(int(task[i]), Motif(text[i * W:(i + 1) * W], int(padding))) for i in range(cnt)]), dtype=np.int64)
            np.ctypeslib.as_array(out, shape=(cnt, 2))[:] = res
            return 0
        except Exception as e:
            err.append(e)
            return -3

    def _window(_user, cnt, task, kind, motifs, out):
        try:
            text = C.string_at(motifs, cnt * W).decode("ascii")
            res = np.asarray(window_fn([(int(task[i]), "remove" if kind[i] else "pssm", Motif(text[i * W:(i + 1) * W], int(padding)))
                                        for i in range(cnt)]), dtype=np.int32)
            np.ctypeslib.as_array(out, shape=(cnt, 2 + 4 * ((W + 63) // 64 * 64)))[:] = res
            return 0
        except Exception as e:
            err.append(e)
            return -3
    params = _params(padding, min_kl, score_threshold, **kw)
    handle = C.c_void_p()
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    rc = lib.nm_search_run_custom(n, C.byref(params), p(pssm, C.c_double), p(totals, C.c_uint64), p(canon, C.c_uint8),
                                  _lib.SEARCH_SCORE_FN(_score), _lib.SEARCH_WINDOW_FN(_window), None, C.byref(handle))
    if err:
        raise err[0]
    _lib.check(rc)
    return SearchResults(lib, handle, keys, mods, [t[2] for t in tasks], padding)
